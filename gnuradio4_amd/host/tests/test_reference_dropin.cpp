// The reference's own block headers, UNMODIFIED, compiled against this host layer and run in a Graph ("existing block source must
// compile unchanged against the new headers").  Container-only: the two files are included from /root/reference where they lie (-I on the command line);
// nothing of them is copied into this repository, and the program is neither built nor run where /root/reference does not exist.
//   blocks/math/include/gnuradio-4.0/math/Math.hpp      MathOpImpl (const ops) + MathOpMultiPortImpl (n-ary), vectors of blocks/math/test/qa_Math.cpp:59-149
//   blocks/math/include/gnuradio-4.0/math/Rotator.hpp   Rotator, blocks/math/test/qa_Rotator.cpp:69-92
#include <cstdio>
#include <iostream>

#include <gnuradio-4.0/math/Math.hpp>    // the reference's file
#include <gnuradio-4.0/math/Rotator.hpp> // the reference's file

namespace t { // a source and a sink in the shape of the reference's testing blocks, written against the same API
template <typename T>
struct Source : gr::Block<Source<T>> {
    gr::PortOut<T> out;
    std::vector<T> values;
    std::size_t    _at = 0;
    GR_MAKE_REFLECTABLE(Source, out);
    gr::work::Result customWork(std::size_t) {
        if (_at == values.size()) { out.buffer->producer_done = true; return {0, 0, gr::work::Status::DONE}; }
        const std::size_t n = std::min(values.size() - _at, out.buffer->free_space());
        auto              s = out.buffer->write_span(n);
        std::copy_n(values.begin() + static_cast<std::ptrdiff_t>(_at), n, s.begin());
        out.buffer->publish(n);
        _at += n;
        return {n, n, gr::work::Status::OK};
    }
};
template <typename T>
struct Sink : gr::Block<Sink<T>> {
    gr::PortIn<T>  in;
    std::vector<T> got;
    GR_MAKE_REFLECTABLE(Sink, in);
    gr::work::Result customWork(std::size_t) {
        const std::size_t n = in.buffer->available();
        if (n == 0) return {0, 0, in.buffer->done() ? gr::work::Status::DONE : gr::work::Status::INSUFFICIENT_INPUT_ITEMS};
        auto s = in.buffer->read_span(n);
        got.insert(got.end(), s.begin(), s.end());
        in.buffer->consume(n);
        return {n, n, gr::work::Status::OK};
    }
};
} // namespace t

static int errors = 0;
template <typename T>
void expect(const char* what, const std::vector<T>& got, const std::vector<T>& want) {
    const bool ok = got == want;
    std::printf("%-58s %s\n", what, ok ? "ok" : "MISMATCH");
    if (!ok) ++errors;
}

template <typename T, template <typename> class BlockT>
std::vector<T> run_const(T value, const std::vector<T>& x) {
    gr::Graph g;
    auto&     src = g.emplaceBlock<t::Source<T>>();
    src.values    = x;
    auto& blk     = g.emplaceBlock<BlockT<T>>(gr::property_map{{"value", static_cast<double>(value)}});
    auto& snk     = g.emplaceBlock<t::Sink<T>>();
    if (!g.connect<"out", "in">(src, blk) || !g.connect<"out", "in">(blk, snk)) ++errors;
    gr::scheduler::Simple sched;
    sched.exchange(std::move(g));
    if (!sched.runAndWait()) ++errors;
    return snk.got;
}
template <typename T, template <typename> class BlockT>
std::vector<T> run_nary(const std::vector<std::vector<T>>& ins) {
    gr::Graph g;
    auto&     blk = g.emplaceBlock<BlockT<T>>(gr::property_map{{"n_inputs", static_cast<std::int64_t>(ins.size())}});
    auto&     snk = g.emplaceBlock<t::Sink<T>>();
    for (std::size_t i = 0; i < ins.size(); ++i) {
        auto& src  = g.emplaceBlock<t::Source<T>>();
        src.values = ins[i];
        if (!g.connect(src, "out", blk, "in#" + std::to_string(i))) ++errors; // runtime variant, vector port "in#i" (Graph.hpp:564-593)
    }
    if (!g.connect<"out", "in">(blk, snk)) ++errors;
    gr::scheduler::Simple sched;
    sched.exchange(std::move(g));
    if (!sched.runAndWait()) ++errors;
    return snk.got;
}

int main() {
    using namespace gr::blocks::math;
    // ---- qa_Math.cpp:123-149: x (op) 2 for the const blocks
    expect<std::int32_t>("AddConst<int32>       (Math.hpp unmodified)", run_const<std::int32_t, AddConst>(2, {1, 2, 8, 17}), {3, 4, 10, 19});
    expect<std::int32_t>("SubtractConst<int32>", run_const<std::int32_t, SubtractConst>(2, {4, 6, 8, 10}), {2, 4, 6, 8});
    expect<float>("MultiplyConst<float>", run_const<float, MultiplyConst>(2.f, {1.f, 2.f, 3.f, 4.f}), {2.f, 4.f, 6.f, 8.f});
    expect<double>("DivideConst<double>", run_const<double, DivideConst>(2.0, {2.0, 4.0, 8.0, 20.0}), {1.0, 2.0, 4.0, 10.0});
    expect<std::uint8_t>("MultiplyConst<uint8> wraps like C++", run_const<std::uint8_t, MultiplyConst>(2, {100, 200}), {200, static_cast<std::uint8_t>(400)});
    // ---- qa_Math.cpp:59-121: n-ary blocks, 1 .. 3 inputs
    expect<std::int32_t>("Add<int32> 3 inputs   (MathOpMultiPortImpl unmodified)", run_nary<std::int32_t, Add>({{1, 2, 3}, {4, 5, 6}, {7, 8, 9}}), {12, 15, 18});
    expect<std::int32_t>("Subtract<int32> 2 inputs", run_nary<std::int32_t, Subtract>({{9, 8, 7}, {1, 2, 3}}), {8, 6, 4});
    expect<float>("Multiply<float> 3 inputs", run_nary<float, Multiply>({{1.f, 2.f}, {3.f, 4.f}, {5.f, 6.f}}), {15.f, 48.f});
    expect<std::int32_t>("Divide<int32> truncates", run_nary<std::int32_t, Divide>({{9, 20}, {2, 3}}), {4, 6});
    expect<double>("Add<double> 1 input", run_nary<double, Add>({{1.5, 2.5}}), {1.5, 2.5});
    // ---- qa_Rotator.cpp:69-92: output[i] angle = (i + 1) pi / 2
    {
        using C = std::complex<float>;
        gr::Graph g;
        auto&     src = g.emplaceBlock<t::Source<C>>();
        src.values.assign(8, C(1.f, 0.f));
        auto& rot = g.emplaceBlock<Rotator<C>>(gr::property_map{{"phase_increment", std::numbers::pi / 2}});
        auto& snk = g.emplaceBlock<t::Sink<C>>();
        if (!g.connect<"out", "in">(src, rot) || !g.connect<"out", "in">(rot, snk)) ++errors;
        const float fs = rot.frequency_shift; // settingsChanged: frequency_shift = phase_increment / (2 pi) * sample_rate
        gr::scheduler::Simple sched;
        sched.exchange(std::move(g));
        if (!sched.runAndWait()) ++errors;
        bool ok = snk.got.size() == 8 && std::abs(fs - 0.25f) < 1e-6f;
        for (std::size_t i = 0; i < snk.got.size(); ++i) {
            const double want = static_cast<double>(i + 1) * std::numbers::pi / 2;
            ok                = ok && std::abs(snk.got[i].real() - std::cos(want)) < 1e-5 && std::abs(snk.got[i].imag() - std::sin(want)) < 1e-5;
        }
        std::printf("%-58s %s\n", "Rotator<complex<float>> angles (Rotator.hpp unmodified)", ok ? "ok" : "MISMATCH");
        if (!ok) ++errors;
        // the XOR rule of settingsChanged throws gr::exception through the layer's applySettings
        bool threw = false;
        try {
            Rotator<C> r2;
            r2.applySettings({{"phase_increment", 0.1}, {"frequency_shift", 0.2}});
        } catch (const gr::exception& e) { threw = std::string(e.what()).find("XOR") != std::string::npos; }
        std::printf("%-58s %s\n", "Rotator: both settings at once -> gr::exception", threw ? "ok" : "MISMATCH");
        if (!threw) ++errors;
    }
    std::printf(errors ? "reference drop-in: %d FAILURES\n" : "reference drop-in: all checks passed\n", errors);
    return errors ? 1 : 0;
}
