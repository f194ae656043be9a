// Host-layer self test (no GPU): the kept gr::Block<>/Port<>/Graph::connect surface and scheduler::Simple, in the style of the
// reference's own tests (blocks/math/test/qa_Math.cpp:16-41, blocks/filter/test/qa_filter.cpp:267-293, core/test/qa_Block.cpp:1315-1343).
// Also BASELINE.json configs[0]: SignalSource -> 64-tap float FIR -> sink, 1 000 448 samples, CPU scheduler; the stream is dumped
// for the python test to compare against the oracle.
#include <deque>
#include <cstdio>
#include <fstream>
#include <iostream>

#include <gr4/blocks.hpp>
#include <gr4/merge.hpp>

using namespace gr;
using namespace std::string_literals;

static int failures = 0;
#define EXPECT(cond)                                                                  \
    do {                                                                              \
        if (!(cond)) { ++failures; std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); } \
    } while (0)

template <typename T, typename BlockUnderTest>
void math_case(const std::vector<std::vector<T>>& inputs, const std::vector<T>& expected) {
    Graph graph;
    auto& block = graph.emplaceBlock<BlockUnderTest>({{"n_inputs", std::int64_t(inputs.size())}});
    for (std::size_t i = 0; i < inputs.size(); ++i) {
        std::vector<std::int64_t> vi; std::vector<double> vd;
        auto& src = graph.emplaceBlock<testing::VectorSource<T>>();
        src.values = inputs[i];
        EXPECT(graph.connect(src, "out"s, block, "in#"s + std::to_string(i)).has_value());
    }
    auto& sink = graph.emplaceBlock<testing::VectorSink<T>>();
    EXPECT(graph.connect(block, "out"s, sink, "in"s).has_value());
    scheduler::Simple sched;
    EXPECT(sched.exchange(std::move(graph)).has_value());
    EXPECT(sched.runAndWait().has_value());
    EXPECT(sink._samples == expected);
}

template <typename T>
void math_suite() { // vectors of qa_Math.cpp:59-121 (integer-valued rows)
    using namespace gr::blocks::math;
    math_case<T, Add<T>>({{1, 2, 8, 17}}, {1, 2, 8, 17});
    math_case<T, Add<T>>({{12, 35, 18, 17}, {31, 15, 27, 36}, {83, 46, 37, 41}}, {126, 96, 82, 94});
    math_case<T, Subtract<T>>({{15, 38, 88, 29}, {3, 12, 26, 18}, {0, 10, 50, 7}}, {12, 16, 12, 4});
    math_case<T, Multiply<T>>({{0, 1, 2, 3}, {4, 5, 6, 2}, {8, 9, 10, 11}}, {0, 45, 120, 66});
    math_case<T, Divide<T>>({{0, 10, 40, 80}, {1, 2, 4, 20}, {1, 5, 5, 2}}, {0, 1, 2, 2});
    AddConst<T> a;      EXPECT(a.processOne(T(4)) == T(5));
    a.applySettings({{"value", std::int64_t(2)}}); EXPECT(a.processOne(T(4)) == T(6));
    DivideConst<T> d;   d.applySettings({{"value", std::int64_t(2)}}); EXPECT(d.processOne(T(4)) == T(2));
}

// gr::UncertainValue<float | double>, two of the fourteen registered math types (Math.hpp:25-28, 68-71): the reference's known answers for its operators
// (meta/test/qa_UncertainValue.cpp:35-42, 98-102, 148-152, 190-194) through the const blocks and the multi-port blocks
template <typename F>
void uncertain_suite() {
    using namespace gr::blocks::math;
    using U = gr::UncertainValue<F>;
    const auto near = [](U got, F v, F u) { return std::abs(got.value - v) <= F(1e-6) * std::abs(v) && std::abs(got.uncertainty - u) <= F(1e-6) * std::abs(u); };
    AddConst<U> a;      a.applySettings({{"value", std::vector<double>{20.0, 4.0}}}); EXPECT(near(a.processOne(U{10, 3}), 30, 5));
    SubtractConst<U> s; s.applySettings({{"value", std::vector<double>{10.0, 4.0}}}); EXPECT(near(s.processOne(U{20, 3}), 10, 5));
    MultiplyConst<U> m; m.applySettings({{"value", std::vector<double>{3.0, 0.5}}});  EXPECT(near(m.processOne(U{4, F(0.5)}), 12, F(2.5)));
    DivideConst<U> d;   d.applySettings({{"value", std::vector<double>{4.0, 0.5}}});  EXPECT(near(d.processOne(U{16, 2}), 4, F(0.70710678118654752)));
    MultiplyConst<U> one;                                                             EXPECT(near(one.processOne(U{4, F(0.5)}), 4, F(0.5))); // default value 1 (no uncertainty)
    one.applySettings({{"value", 3.0}});                                              EXPECT(near(one.processOne(U{4, F(0.5)}), 12, F(1.5))); // a plain number: exact
    math_case<U, Add<U>>({{U{10, 3}, U{1, 0}}, {U{20, 4}, U{2, 0}}}, {U{30, 5}, U{3, 0}});
    math_case<U, Multiply<U>>({{U{4, F(0.5)}}, {U{3, F(0.5)}}}, {U{12, F(2.5)}});
    EXPECT(gr::value(U{3, 1}) == F(3) && gr::uncertainty(U{3, 1}) == F(1) && gr::uncertainty(F(42)) == F(0) && !gr::UncertainValueLike<F> && gr::UncertainValueLike<U>);
}

int main(int argc, char** argv) {
    // ---- math blocks through Graph + Scheduler for the integer and float types
    math_suite<std::uint8_t>(); math_suite<std::int16_t>(); math_suite<std::int32_t>(); math_suite<std::uint64_t>(); math_suite<float>(); math_suite<double>();
    uncertain_suite<float>(); uncertain_suite<double>();

    // ---- connect(): errors are returned, not thrown (docs/USER_API_Connecting_Blocks.md "Error handling")
    {
        Graph g;
        auto& s = g.emplaceBlock<testing::VectorSource<float>>();
        auto& k = g.emplaceBlock<testing::VectorSink<double>>();
        auto& f = g.emplaceBlock<filter::fir_filter<float>>();
        EXPECT(!(g.connect<"out", "in">(s, k)).has_value());          // value_type mismatch
        EXPECT(!(g.connect(s, "nope"s, f, "in"s)).has_value());        // unknown port
        EXPECT(!(g.connect(s, "out"s, f, "out"s)).has_value());        // wrong direction
        EXPECT((g.connect<"out", "in">(s, f)).has_value());
        bool threw = false;
        try { f.applySettings({{"no_such_setting", 1.0}}); } catch (const std::invalid_argument&) { threw = true; }
        EXPECT(threw);
    }
    // ---- edge storage handed to a neighbour that moves the samples itself (a copy engine, copy threads): lend / consume on the reading side,
    //      reserve / publish on the writing side; the storage must not move while spans are out, and readers only see published items
    {
        EdgeBuffer<int> e(8); // capacity 8, storage 16
        EdgeBufferBase& b = e;
        for (int i = 0; i < 8; ++i) e.write_span(1)[0] = i, e.publish(1);
        const int* a = static_cast<const int*>(b.lend_items(3));
        const int* c = static_cast<const int*>(b.lend_items(4));
        EXPECT(a && c && a[0] == 0 && c[0] == 3 && c == a + 3);
        EXPECT(b.available_items() == 1 && b.lend_items(2) == nullptr); // lent items are spoken for; only one is left
        EXPECT(e.free_space() == 0);                                     // still 8 unread items in a capacity of 8
        b.consume_items(3);                                              // the oldest span comes back
        EXPECT(e.read_pos == 3 && e.free_space() == 3 && b.available_items() == 1);
        for (int i = 8; i < 11; ++i) e.write_span(1)[0] = i, e.publish(1);
        EXPECT(c[0] == 3 && c[3] == 6);                                  // the span still out has not moved
        EXPECT(e.free_space() == 0);
        b.consume_items(4);
        EXPECT(e.free_space() == 4 && e.lent == 0);
        // the tail has reached 11 of 16: with a span lent the writer only gets the contiguous room that is left, afterwards the storage compacts again
        const int* d = static_cast<const int*>(b.lend_items(4));
        EXPECT(d && d[0] == 7 && e.free_space() == 4);
        for (int i = 11; i < 15; ++i) e.write_span(1)[0] = i, e.publish(1);
        EXPECT(e.tail == 15 && e.free_space() == 0);
        b.consume_items(4);
        EXPECT(e.free_space() == 4);
        e.write_span(4)[0] = 15; // compacts: 4 unread items move to the front
        EXPECT(e.head == 0 && e.tail == 4 && e.read_span(4)[0] == 11 && e.data[4] == 15);
        e.publish(1);
        // writing side
        int* r1 = static_cast<int*>(b.reserve_items(2));
        int* r2 = static_cast<int*>(b.reserve_items(1));
        EXPECT(r1 && r2 == r1 + 2 && e.available() == 5 && e.free_space() == 0 && b.reserve_items(1) == nullptr);
        r1[0] = 100; r1[1] = 101; r2[0] = 102;
        b.publish_reserved(2);
        EXPECT(e.available() == 7 && e.read_span(7)[5] == 100 && e.reserved == 1);
        b.publish_reserved(1);
        EXPECT(e.available() == 8 && e.read_span(8)[7] == 102 && e.write_pos == 19);
    }
    // ---- the same edge under 200 000 random operations against a plain queue: typed writes, lends / releases in order, reservations / publications in order;
    //      spans handed out must stay where they are and hold the right items until they come back
    {
        EdgeBuffer<int>  e(64);
        EdgeBufferBase&  b = e;
        std::deque<int>  model;                                    // published and not yet consumed
        std::deque<std::pair<const int*, std::size_t>> lent;       // outstanding lent spans (pointer, n) with their first value = model index base
        std::deque<std::pair<int*, std::size_t>>       reserved;   // outstanding reservations
        std::deque<int>                                reserved_first; // first value each reservation will hold
        std::size_t   lent_items = 0;
        int           next = 0;
        std::uint32_t rng = 2463534242u;
        const auto    rnd = [&](std::uint32_t m) { rng ^= rng << 13; rng ^= rng >> 17; rng ^= rng << 5; return rng % m; };
        bool ok = true;
        for (int step = 0; step < 200000 && ok; ++step) {
            switch (rnd(6)) {
            case 0: { // typed write (only while nothing is reserved: one writer per edge)
                if (!reserved.empty()) break;
                const std::size_t n = std::min<std::size_t>(e.free_space(), rnd(40));
                if (n == 0) break;
                auto sp = e.write_span(n);
                for (std::size_t i = 0; i < n; ++i) { sp[i] = next; model.push_back(next++); }
                e.publish(n);
                break;
            }
            case 1: { // lend
                const std::size_t n = rnd(30) + 1;
                const int* p = static_cast<const int*>(b.lend_items(n));
                if (n > model.size() - lent_items) { ok = ok && p == nullptr; break; }
                ok = ok && p != nullptr && p[0] == model[lent_items] && p[n - 1] == model[lent_items + n - 1];
                lent.emplace_back(p, n);
                lent_items += n;
                break;
            }
            case 2: { // give the oldest lent span back: it still holds what it held
                if (lent.empty()) break;
                auto [p, n] = lent.front();
                ok = ok && p[0] == model.front() && p[n - 1] == model[n - 1];
                b.consume_items(n);
                model.erase(model.begin(), model.begin() + static_cast<std::ptrdiff_t>(n));
                lent.pop_front();
                lent_items -= n;
                break;
            }
            case 3: { // typed consume (only while nothing is lent)
                if (!lent.empty() || model.empty()) break;
                const std::size_t n = rnd(static_cast<std::uint32_t>(model.size())) + 1;
                auto sp = e.read_span(n);
                ok = ok && sp[0] == model.front() && sp[n - 1] == model[n - 1];
                e.consume(n);
                model.erase(model.begin(), model.begin() + static_cast<std::ptrdiff_t>(n));
                break;
            }
            case 4: { // reserve
                const std::size_t n = rnd(30) + 1;
                int* p = static_cast<int*>(b.reserve_items(n));
                if (!p) break;
                for (std::size_t i = 0; i < n; ++i) p[i] = next + static_cast<int>(i);
                reserved.emplace_back(p, n);
                reserved_first.push_back(next);
                next += static_cast<int>(n);
                break;
            }
            case 5: { // publish the oldest reservation: it has not moved
                if (reserved.empty()) break;
                auto [p, n] = reserved.front();
                ok = ok && p[0] == reserved_first.front() && p[n - 1] == reserved_first.front() + static_cast<int>(n) - 1;
                b.publish_reserved(n);
                for (std::size_t i = 0; i < n; ++i) model.push_back(reserved_first.front() + static_cast<int>(i));
                reserved.pop_front();
                reserved_first.pop_front();
                break;
            }
            }
            ok = ok && e.available() == model.size() && b.available_items() == model.size() - lent_items && e.available() + e.reserved <= e.capacity;
        }
        EXPECT(ok);
    }
    // ---- memory seam: ComputeRegistry providers and per-edge resources (ComputeDomain.hpp:105-173, Graph.hpp:738-775)
    {
        struct Counting final : std::pmr::memory_resource {
            std::size_t bytes = 0;
            void* do_allocate(std::size_t n, std::size_t a) override { bytes += n; return std::pmr::new_delete_resource()->allocate(n, a); }
            void  do_deallocate(void* p, std::size_t n, std::size_t a) override { std::pmr::new_delete_resource()->deallocate(p, n, a); }
            bool  do_is_equal(const std::pmr::memory_resource& o) const noexcept override { return this == &o; }
        };
        static Counting counting;
        auto& reg = ComputeRegistry::instance();
        EXPECT(reg.resolve(ComputeDomain::parse("host")).value() == std::pmr::new_delete_resource());
        EXPECT(!reg.resolve(ComputeDomain::parse("gpu:fake:0")).has_value() && reg.tryResolve(ComputeDomain::parse("gpu:fake:0")) == nullptr);
        reg.register_provider("fake", [](const ComputeDomain& d, void*) -> std::pmr::memory_resource* { return d.index == 0 ? &counting : nullptr; });
        EXPECT(reg.resolve(ComputeDomain::parse("gpu:fake:0")).value() == &counting);
        EXPECT(!reg.resolve(ComputeDomain::parse("gpu:fake:1")).has_value()); // provider returned null
        Graph g;
        auto& s = g.emplaceBlock<testing::VectorSource<float>>();
        auto& f = g.emplaceBlock<filter::fir_filter<float>>();
        auto& k = g.emplaceBlock<testing::NullSink<float>>();
        EXPECT((g.connect<"out", "in">(s, f, EdgeParameters{.minBufferSize = 1000, .domain = "gpu:fake:0"})).has_value()); // storage from the domain's provider
        EXPECT(f.in.buffer->resource() == &counting && counting.bytes >= 2 * 65536 * sizeof(float));
        EXPECT((g.connect<"out", "in">(f, k, EdgeParameters{.domain = "gpu:fake:1"})).has_value());                         // unresolved domain: default resource
        EXPECT(k.in.buffer->resource() == std::pmr::get_default_resource());
    }
    // ---- Decimator 100 -> 10 samples (qa_filter.cpp:267-293)
    {
        Graph g;
        auto& src = g.emplaceBlock<testing::VectorSource<float>>({{"n_samples_max", std::int64_t(100)}});
        src.values = {1.f, 2.f, 3.f};
        auto& dec  = g.emplaceBlock<filter::Decimator<float>>({{"decim", std::int64_t(10)}});
        auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
        EXPECT((g.connect<"out", "in">(src, dec)).has_value());
        EXPECT((g.connect<"out", "in">(dec, sink)).has_value());
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        EXPECT(sched.runAndWait().has_value());
        EXPECT(dec.input_chunk_size == 10u && dec.output_chunk_size == 1u);
        EXPECT(sink._samples.size() == 10u);
        EXPECT(sink._samples[1] == src.values[10 % 3]);
    }
    // ---- tags: forwarded at the first output sample of a chunk, "gr:" keys only, gr:sample_rate scaled by the resampling ratio (qa_filter.cpp:267-293, 318-320);
    //      chunks split at tags; settings-by-tag (Settings.hpp:433)
    {
        Graph g;
        auto& src = g.emplaceBlock<testing::VectorSource<float>>({{"n_samples_max", std::int64_t(100)}});
        src.values = {1.f, 2.f, 3.f};
        src._tags  = {{0, {{"gr:sample_rate", 1000.f}}}, {50, {{"gr:sample_rate", 2000.f}, {"custom", std::int64_t(1)}}}};
        auto& dec  = g.emplaceBlock<filter::Decimator<float>>({{"decim", std::int64_t(10)}});
        auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
        g.connect<"out", "in">(src, dec);
        g.connect<"out", "in">(dec, sink);
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        EXPECT(sched.runAndWait().has_value());
        EXPECT(sink._samples.size() == 10u && sink._tags.size() == 2u);
        if (sink._tags.size() == 2) {
            EXPECT(sink._tags[0].index == 0u && (sink._tags[0].map == property_map{{"gr:sample_rate", 100.f}}));
            EXPECT(sink._tags[1].index == 5u && (sink._tags[1].map == property_map{{"gr:sample_rate", 200.f}})); // "custom" has no gr: prefix: not forwarded
        }
    }
    { // tags INSIDE a chunk that cannot be split (a decimation group of 100): all tags of the chunk are merged, later ones overriding earlier ones,
      // and forwarded at the chunk's first output sample -- none is lost (mergedInputTag, Block.hpp:1511-1530)
        Graph g;
        auto& src = g.emplaceBlock<testing::VectorSource<float>>({{"n_samples_max", std::int64_t(500)}});
        src.values = {1.f, 2.f, 3.f};
        src._tags  = {{0, {{"gr:sample_rate", 1000.f}}}, {250, {{"gr:sample_rate", 2000.f}, {"gr:trigger_name", "a"s}}}, {260, {{"gr:trigger_name", "b"s}}}, {400, {{"gr:trigger_name", "c"s}}}};
        auto& dec  = g.emplaceBlock<filter::Decimator<float>>({{"decim", std::int64_t(100)}});
        auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
        g.connect<"out", "in">(src, dec);
        g.connect<"out", "in">(dec, sink);
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        EXPECT(sched.runAndWait().has_value());
        EXPECT(sink._samples.size() == 5u && sink._tags.size() == 3u);
        if (sink._tags.size() == 3) {
            EXPECT(sink._tags[0].index == 0u && (sink._tags[0].map == property_map{{"gr:sample_rate", 10.f}}));
            EXPECT(sink._tags[1].index == 2u && (sink._tags[1].map == property_map{{"gr:sample_rate", 20.f}, {"gr:trigger_name", "b"s}})); // samples 200..299: tags at 250 and 260
            EXPECT(sink._tags[2].index == 4u && (sink._tags[2].map == property_map{{"gr:trigger_name", "c"s}}));
        }
    }
    {
        Graph g;
        auto& src = g.emplaceBlock<testing::VectorSource<float>>({{"n_samples_max", std::int64_t(100000)}});
        src.values = {1.f};
        src._tags  = {{70001, {{"value", 3.0}, {"gr:trigger_name", "go"s}}}, {70002, {{"gr:value", 5.0}}}};
        auto& mul  = g.emplaceBlock<blocks::math::MultiplyConst<float>>({{"value", 2.0}});
        auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
        g.connect<"out", "in">(src, mul);
        g.connect<"out", "in">(mul, sink);
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        EXPECT(sched.runAndWait().has_value());
        bool ok = sink._samples.size() == 100000u;
        for (std::size_t i = 0; ok && i < sink._samples.size(); ++i) ok = sink._samples[i] == (i < 70001 ? 2.f : i < 70002 ? 3.f : 5.f); // the change lands exactly on the tagged sample
        EXPECT(ok);
        EXPECT(mul._settings_by_tag == 2u && mul.value == 5.f);
        EXPECT(sink._tags.size() == 2u && sink._tags[0].index == 70001u && (sink._tags[0].map == property_map{{"gr:trigger_name", "go"s}}) && sink._tags[1].index == 70002u);
    }
    // ---- merge API: the reference benchmark's IIR low-pass y[n] = a x[n] + (1 - a) y[n-1] as Merge<MultiplyConst, FeedbackMerge<Adder, MultiplyConst>>
    //      (core/benchmarks/bm_MergeApi.cpp:59-77), type spelled as upstream; sub-block settings by dotted keys (shorthand) and as upstream's nested maps
    {
        using blocks::math::MultiplyConst;
        using IIRChain = gr::Merge<MultiplyConst<float>, "out", gr::FeedbackMerge<Adder<>, "out", MultiplyConst<float>, "out", "in2">, "in1">;
        constexpr float kAlpha = 0.3f;
        Graph g;
        auto& src = g.emplaceBlock<testing::VectorSource<float>>({{"n_samples_max", std::int64_t(5000)}});
        src.values = {1.f, -0.5f, 0.25f, 2.f, 0.f, 0.f, -3.f};
        auto& iir  = g.emplaceBlock<IIRChain>({{"leftBlock.value", double(kAlpha)}, {"rightBlock.feedback.value", double(1.0f - kAlpha)}});
        auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
        EXPECT((g.connect<"out", "in">(src, iir)).has_value() && (g.connect<"out", "in">(iir, sink)).has_value());
        EXPECT(iir.leftBlock.value == kAlpha && iir.rightBlock.feedback.value == 1.0f - kAlpha);
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        EXPECT(sched.runAndWait().has_value());
        float state = 0.f, worst = 0.f; // IIRReference of the benchmark
        for (std::size_t i = 0; i < sink._samples.size(); ++i) {
            state = kAlpha * src.values[i % src.values.size()] + (1.0f - kAlpha) * state;
            worst = std::max(worst, std::abs(sink._samples[i] - state));
        }
        EXPECT(sink._samples.size() == 5000u && worst <= 1e-6f);
        bool threw = false;
        try { IIRChain bad; bad.applySettings({{"middleBlock.value", 1.0}}); } catch (const std::invalid_argument&) { threw = true; }
        EXPECT(threw);
        // settings as upstream spells them (BlockMerging.hpp:93-110, 206-210, 641-649): nested maps under the part's name; a flat key goes to every part that has it
        {
            IIRChain nested;
            nested.applySettings({{"leftBlock", property_map{{"value", double(kAlpha)}}}, {"rightBlock", property_map{{"feedback", property_map{{"value", double(1.0f - kAlpha)}}}}}});
            EXPECT(nested.leftBlock.value == kAlpha && nested.rightBlock.feedback.value == 1.0f - kAlpha);
            IIRChain flat;
            flat.applySettings({{"value", 0.5}, {"name", "lp"s}}); // both MultiplyConst parts have `value`; the name is the merged block's own
            EXPECT(flat.leftBlock.value == 0.5f && flat.rightBlock.feedback.value == 0.5f && flat.name == "lp");
            bool bad_part = false;
            try { flat.applySettings({{"middleBlock", property_map{{"value", 1.0}}}}); } catch (const std::invalid_argument&) { bad_part = true; }
            EXPECT(bad_part);
            // the exposed input of a FeedbackMerge under the name the forward block gave it (the adder's free `in1`), as upstream's graphs address it
            using Loop = gr::FeedbackMerge<Adder<>, "out", MultiplyConst<float>, "out", "in2">;
            Graph g2;
            auto& s2 = g2.emplaceBlock<testing::VectorSource<float>>({{"n_samples_max", std::int64_t(4)}});
            s2.values = {1.f, 1.f, 1.f, 1.f};
            auto& lp  = g2.emplaceBlock<Loop>({{"feedback", property_map{{"value", 0.5}}}});
            auto& k2  = g2.emplaceBlock<testing::VectorSink<float>>();
            EXPECT((g2.connect<"out", "in1">(s2, lp)).has_value() && (g2.connect<"out", "in">(lp, k2)).has_value());
            EXPECT(!(g2.connect<"out", "in2">(s2, lp)).has_value()); // the feedback input is closed inside the block
            scheduler::Simple sc2;
            sc2.exchange(std::move(g2));
            EXPECT(sc2.runAndWait().has_value());
            EXPECT((k2._samples == std::vector<float>{1.f, 1.5f, 1.75f, 1.875f}));
        }
        // SplitMergeCombine: fan-out to N paths, signed sum (USER_API_Connecting_Blocks.md "SplitMergeCombine"; bm_MergeApi.cpp:62-67)
        using FanOut = gr::SplitMergeCombine<MultiplyConst<float>, MultiplyConst<float>>;
        FanOut fan;
        fan.applySettings({{"path0.value", 2.0}, {"path1.value", 3.0}});
        EXPECT(fan.processOne(1.5f) == 7.5f && fan.path<1>().value == 3.f);
        using Diff = gr::SplitMergeCombine<gr::OutputSigns<+1.0f, -1.0f>, MultiplyConst<float>, MultiplyConst<float>>;
        Diff diff;
        diff.applySettings({{"path0.value", 2.0}, {"path1.value", 3.0}});
        EXPECT(diff.processOne(2.f) == -2.f);
        threw = false;
        try { diff.applySettings({{"path2.value", 1.0}}); } catch (const std::invalid_argument&) { threw = true; }
        EXPECT(threw);
        // the IIR low-pass with its feedback gain decomposed: feedback = y + (-a) y = (1 - a) y
        using IIRChainSplitMerge = gr::Merge<MultiplyConst<float>, "out", gr::FeedbackMerge<Adder<>, "out", gr::SplitMergeCombine<MultiplyConst<float>, MultiplyConst<float>>, "out", "in2">, "in1">;
        IIRChainSplitMerge sm;
        sm.applySettings({{"leftBlock.value", double(kAlpha)}, {"rightBlock.feedback.path0.value", 1.0}, {"rightBlock.feedback.path1.value", double(-kAlpha)}});
        float st2 = 0.f, w2 = 0.f;
        for (int i = 0; i < 2000; ++i) {
            const float xin = src.values[static_cast<std::size_t>(i) % src.values.size()];
            st2 = kAlpha * xin + (1.0f - kAlpha) * st2;
            w2  = std::max(w2, std::abs(sm.processOne(xin) - st2));
        }
        EXPECT(w2 <= 1e-6f);
    }
    // ---- fan-out: one output port wired to two inputs (Graph.hpp:595-690 allows any number of readers per output); samples and tags reach both
    {
        Graph g;
        auto& src = g.emplaceBlock<testing::VectorSource<float>>({{"n_samples_max", std::int64_t(200000)}});
        src.values = {1.f, 2.f, 3.f};
        src._tags  = {{70000, {{"gr:trigger_name", "t0"s}}}};
        auto& dbl  = g.emplaceBlock<blocks::math::MultiplyConst<float>>({{"value", 2.0}});
        auto& inc  = g.emplaceBlock<blocks::math::AddConst<float>>({{"value", 1.0}});
        auto& dec  = g.emplaceBlock<filter::Decimator<float>>({{"decim", std::int64_t(4)}});
        auto& s1 = g.emplaceBlock<testing::VectorSink<float>>();
        auto& s2 = g.emplaceBlock<testing::VectorSink<float>>();
        auto& s3 = g.emplaceBlock<testing::VectorSink<float>>();
        EXPECT((g.connect<"out", "in">(src, dbl)).has_value());
        EXPECT((g.connect<"out", "in">(dbl, s1)).has_value());  // first reader of dbl.out
        EXPECT((g.connect<"out", "in">(dbl, inc)).has_value()); // second reader: a mirror buffer
        EXPECT((g.connect<"out", "in">(dbl, dec)).has_value()); // third reader, at another rate
        EXPECT((g.connect<"out", "in">(inc, s2)).has_value());
        EXPECT((g.connect<"out", "in">(dec, s3)).has_value());
        EXPECT(dbl.out.buffer->mirrors.size() == 2u && s1.in.buffer == dbl.out.buffer && inc.in.buffer == dbl.out.buffer->mirrors[0]);
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        EXPECT(sched.runAndWait().has_value());
        bool ok = s1._samples.size() == 200000u && s2._samples.size() == 200000u && s3._samples.size() == 50000u;
        for (std::size_t i = 0; ok && i < 200000; ++i) ok = s1._samples[i] == 2.f * src.values[i % 3] && s2._samples[i] == 2.f * src.values[i % 3] + 1.f;
        for (std::size_t i = 0; ok && i < 50000; ++i) ok = s3._samples[i] == 2.f * src.values[(4 * i) % 3];
        EXPECT(ok);
        EXPECT(s1._tags.size() == 1u && s2._tags.size() == 1u && s3._tags.size() == 1u);
        if (s1._tags.size() == 1 && s2._tags.size() == 1 && s3._tags.size() == 1)
            EXPECT(s1._tags[0].index == 70000u && s2._tags[0].index == 70000u && s3._tags[0].index == 17500u && s2._tags[0].map == s1._tags[0].map);
    }
    // ---- FIR box-car step response settles in 10 samples; IIR forms agree (qa_filter.cpp:53-128)
    {
        filter::fir_filter<double> fir;
        fir.applySettings({{"b", std::vector<double>(10, 0.1)}});
        double last = 0;
        for (int i = 0; i < 20; ++i) { last = fir.processOne(i == 0 ? 0.0 : 1.0); if (i == 0) EXPECT(last == 0.0); }
        EXPECT(std::abs(last - 1.0) < 1e-12);
        const std::vector<double> b{0.020083365564211, 0.040166731128423, 0.020083365564211}, a{1.0, -1.561018075800718, 0.641351538057563};
        filter::iir_filter<double, filter::IIRForm::DF_I> f1; filter::iir_filter<double, filter::IIRForm::DF_II> f2;
        filter::iir_filter<double, filter::IIRForm::DF_I_TRANSPOSED> f3; filter::iir_filter<double, filter::IIRForm::DF_II_TRANSPOSED> f4;
        for (auto* f : {static_cast<void*>(&f1)}) (void)f;
        f1.applySettings({{"b", b}, {"a", a}}); f2.applySettings({{"b", b}, {"a", a}}); f3.applySettings({{"b", b}, {"a", a}}); f4.applySettings({{"b", b}, {"a", a}});
        for (int i = 0; i < 20; ++i) {
            const double x = i == 0 ? 0.0 : 1.0, y1 = f1.processOne(x);
            EXPECT(std::abs(f2.processOne(x) - y1) < 1e-5 && std::abs(f3.processOne(x) - y1) < 1e-5 && std::abs(f4.processOne(x) - y1) < 1e-5);
        }
    }
    // ---- rotator: output[i] angle = (i+1) * pi/2 (qa_Rotator.cpp:69-92); XOR settings
    {
        blocks::math::Rotator<std::complex<float>> rot;
        rot.applySettings({{"phase_increment", double(std::numbers::pi / 2)}});
        EXPECT(std::abs(rot.frequency_shift - 0.25f) < 1e-3f);
        for (int i = 0; i < 8; ++i) {
            const auto y = rot.processOne({1.f, 0.f});
            EXPECT(std::abs(y.real() - std::cos((i + 1) * std::numbers::pi / 2)) < 1e-5 && std::abs(y.imag() - std::sin((i + 1) * std::numbers::pi / 2)) < 1e-5);
        }
        bool threw = false;
        try { rot.applySettings({{"phase_increment", 0.1}, {"frequency_shift", 0.2}}); } catch (const std::invalid_argument&) { threw = true; }
        EXPECT(threw);
    }
    // ---- the inert-seam behaviour the reference pins (qa_Block.cpp:1315-1343): a block without a device kernel warns once, runs on host
    {
        Graph g;
        auto& src = g.emplaceBlock<testing::VectorSource<float>>({{"n_samples_max", std::int64_t(5000)}});
        src.values = {1.f};
        auto& dec  = g.emplaceBlock<filter::Decimator<float>>({{"decim", std::int64_t(5)}, {"compute_domain", "gpu:hip:0"s}});
        int   warnings = 0;
        dec._log = [&](std::string_view) { ++warnings; };
        auto& sink = g.emplaceBlock<testing::NullSink<float>>();
        g.connect<"out", "in">(src, dec);
        g.connect<"out", "in">(dec, sink);
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        EXPECT(sched.runAndWait().has_value());
        EXPECT(sink._count == 1000u && warnings == 1);
        const auto d = ComputeDomain::parse("gpu:hip:1");
        EXPECT(d.kind == "gpu" && d.backend == "hip" && d.index == 1 && d.is_device() && !ComputeDomain::parse("host").is_device());
    }
    // ---- BasicFilter / BasicDecimatingFilter: designed low-pass (order 4, f_low 100 Hz, fs 1 kHz; Hamming FIR / Chebyshev-1 IIR) passes 50 Hz >= 0.9,
    //      attenuates 300 Hz <= 0.2 (qa_filter.cpp:144-265); settings by enumerator name like the reference's property_map strings
    for (const char* kind : {"FIR", "IIR"}) {
        const property_map cfg{{"filter_type", std::string(kind)}, {"filter_response", "LOWPASS"s}, {"filter_order", std::int64_t(4)}, {"f_low", 100.0}, {"sample_rate", 1000.0},
                               {"iir_design_method", "Chebyshev1"s}, {"fir_design_method", "Hamming"s}};
        for (const double freq : {50.0, 300.0}) {
            filter::BasicFilter<float> f;
            f.applySettings(cfg);
            EXPECT((f.filter_type == (std::string(kind) == "FIR" ? filter::FilterType::FIR : filter::FilterType::IIR)));
            float  peak = 0.f;
            double phase = 0;
            for (int i = 0; i < 2000; ++i) {
                phase += 2 * std::numbers::pi * freq / 1000.0;
                const float y = f.processOne(static_cast<float>(std::sin(phase)));
                if (i >= 1000) peak = std::max(peak, std::abs(y)); // ignore the initial transient
            }
            EXPECT(freq < 100 ? peak >= 0.9f : peak <= 0.2f);

            filter::BasicDecimatingFilter<double> d;
            auto dcfg = cfg;
            dcfg["decimate"] = std::int64_t(5);
            d.applySettings(dcfg);
            EXPECT(d.input_chunk_size == 5u);
            std::vector<double> in(1000), out(200);
            double              ph = 0, dpeak = 0;
            for (int rep = 0; rep < 2; ++rep) {
                for (auto& v : in) { ph += 2 * std::numbers::pi * freq / 1000.0; v = std::sin(ph); }
                EXPECT(d.processBulk(in, out) == work::Status::OK);
            }
            for (double v : out) dpeak = std::max(dpeak, std::abs(v));
            EXPECT(freq < 100 ? dpeak >= 0.9 : dpeak <= 0.2);
        }
    }
    {
        filter::BasicFilter<float> f;
        bool threw = false;
        try { f.applySettings({{"filter_type", "NOPE"s}}); } catch (const std::invalid_argument&) { threw = true; }
        EXPECT(threw); // unknown enumerator name
    }
    // ---- FFT block: DataSet per frame; peak at 0.1 fs +- 1/N for N = 256, ranges = min/max of each signal (qa_fourier.cpp:53-109); real input: N/2 bins
    {
        constexpr std::size_t N = 256;
        Graph g;
        auto& src = g.emplaceBlock<testing::VectorSource<std::complex<float>>>();
        for (std::size_t i = 0; i < 3 * N + 17; ++i) src.values.emplace_back(static_cast<float>(std::cos(2 * std::numbers::pi * 0.1 * double(i))), static_cast<float>(std::sin(2 * std::numbers::pi * 0.1 * double(i))));
        auto& fft  = g.emplaceBlock<blocks::fft::FFT<std::complex<float>>>({{"fftSize", std::int64_t(N)}, {"sample_rate", 1.0}, {"window", "hann"s}});
        auto& sink = g.emplaceBlock<testing::VectorSink<DataSet<float>>>();
        EXPECT((g.connect<"out", "in">(src, fft)).has_value());
        EXPECT((g.connect<"out", "in">(fft, sink)).has_value());
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        EXPECT(sched.runAndWait().has_value());
        EXPECT(sink._samples.size() == 3u); // whole frames only, the 17 trailing samples are dropped
        EXPECT(fft._windowType == algorithm::window::Type::Hann && fft.input_chunk_size == N && fft.output_chunk_size == 1u);
        for (const auto& ds : sink._samples) {
            EXPECT(ds.extents == std::vector<std::int32_t>{std::int32_t(N)} && ds.size() == 4u && ds.signal_values.size() == 4 * N && ds.axis_values[0].size() == N);
            const auto        mag  = ds.signalValues(0);
            const std::size_t peak = static_cast<std::size_t>(std::max_element(mag.begin(), mag.end()) - mag.begin());
            EXPECT(std::abs(ds.axisValues(0)[peak] - 0.1f) <= 1.f / N);
            EXPECT(ds.axis_values[0].front() == -0.5f);
            for (std::size_t i = 0; i < 4; ++i) {
                const auto sv = ds.signalValues(i);
                EXPECT(ds.signal_ranges[i].min == *std::min_element(sv.begin(), sv.end()) && ds.signal_ranges[i].max == *std::max_element(sv.begin(), sv.end()));
            }
            EXPECT(ds.signal_names[0] == "Magnitude(unknown signal)" && ds.signal_quantities[3] == "Im(FFT)" && ds.meta_information.size() == 4u);
        }
        blocks::fft::FFT<float> rf;
        rf.applySettings({{"fftSize", std::int64_t(64)}, {"window", "None"s}, {"sample_rate", 64.0}});
        std::vector<float> x(64);
        for (std::size_t i = 0; i < 64; ++i) x[i] = 5.f * static_cast<float>(std::sin(2 * std::numbers::pi * 8.0 * double(i) / 64.0));
        std::vector<DataSet<float>> o(1);
        EXPECT(rf.processBulk(x, o) == work::Status::OK);
        EXPECT(o[0].extents[0] == 32 && o[0].axis_values[0][8] == 8.f);
        EXPECT(std::abs(o[0].signalValues(0)[8] - 5.f) < 1e-4f); // amplitude of the sine at its bin (qa_algorithm_fourier.cpp:78-94)
    }
    // ---- BASELINE configs[0]: SignalSource -> 64-tap float FIR -> sink, 1 000 448 samples (round_up(1e6, 1024), bm_MergeApi.cpp:20)
    {
        constexpr std::size_t N = 1000448, K = 64;
        std::vector<double> taps(K);
        double sum = 0;
        for (std::size_t i = 0; i < K; ++i) { // Hamming windowed-sinc, fc = 0.1, DC gain 1
            const double w = 0.53836 - 0.46164 * std::cos(2 * std::numbers::pi * double(i) / double(K - 1)), x = 0.2 * (double(i) - (K - 1) / 2.0);
            taps[i] = w * 0.2 * (x == 0 ? 1.0 : std::sin(std::numbers::pi * x) / (std::numbers::pi * x));
            sum += taps[i];
        }
        for (auto& t : taps) t /= sum;
        Graph g;
        auto& src  = g.emplaceBlock<basic::SignalGenerator<float>>({{"signal_type", "Sin"s}, {"frequency", 50.0}, {"sample_rate", 1000.0}, {"n_samples_max", std::int64_t(N)}});
        auto& fir  = g.emplaceBlock<filter::fir_filter<float>>({{"b", taps}});
        auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
        EXPECT((g.connect<"out", "in">(src, fir)).has_value());
        EXPECT((g.connect<"out", "in">(fir, sink)).has_value());
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        EXPECT(sched.runAndWait().has_value());
        EXPECT(sink._samples.size() == N); // sample count in == out
        if (argc > 1) {
            std::ofstream f(argv[1], std::ios::binary);
            std::vector<float> tf(taps.begin(), taps.end());
            const std::uint64_t hdr[2] = {K, sink._samples.size()};
            f.write(reinterpret_cast<const char*>(hdr), sizeof hdr);
            f.write(reinterpret_cast<const char*>(tf.data()), K * sizeof(float));
            f.write(reinterpret_cast<const char*>(sink._samples.data()), sink._samples.size() * sizeof(float));
        }
    }
    std::printf(failures ? "host-cpu: %d FAILURES\n" : "host-cpu: all checks passed\n", failures);
    return failures ? 1 : 0;
}
