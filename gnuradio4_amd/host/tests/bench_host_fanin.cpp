// Host-fed rate of the sharded 8-channel graph through the C++ graph API on ONE device (developer tool):
//   C x ( DmaSource<complex<float>> -> fir_filter (gpu:hip:0) -> PowerSpectrum (gpu:hip:0) ) -> Add<float> -> NullSink<float>,  planned into one FanInRun.
// Everything a sample goes through is timed: source loops, host edges, H2D of every channel, the one launch per exchange, D2H of the sum, sink.
//   bench_host_fanin [log2_samples_per_channel = 24] [channels = 8] [frames_per_exchange = 0: what the edges hold] [pinned | pageable]
#include <chrono>
#include <cstdio>
#include <cstdlib>

#include <gr4/hip.hpp>

using namespace gr;
using namespace std::string_literals;

// publishes samples that are already in its port buffer (an SDR driver's DMA target); the storage is filled once at start
template <typename T>
struct DmaSource : Block<DmaSource<T>> {
    PortOut<T>  out;
    Size_t      n_samples_max = 0;
    std::size_t _produced = 0;
    GR_MAKE_REFLECTABLE(DmaSource, out, n_samples_max);
    void fill(std::size_t c) {
        for (std::size_t i = 0, nfill = out.buffer->is_ring() ? out.buffer->capacity : out.buffer->data.size(); i < nfill; ++i) out.buffer->base()[i] = {static_cast<float>((i + c) % 17) - 8.f, static_cast<float>(i % 5)};
    }
    work::Result customWork(std::size_t requested) {
        if (_produced >= n_samples_max) return {requested, 0, work::Status::DONE};
        const std::size_t n = std::min({std::size_t(n_samples_max) - _produced, out.buffer->free_space(), requested});
        if (n == 0) return {requested, 0, work::Status::INSUFFICIENT_OUTPUT_ITEMS};
        (void)out.buffer->write_span(n);
        out.buffer->publish(n);
        _produced += n;
        return {requested, n, work::Status::OK};
    }
};

int main(int argc, char** argv) {
    const std::size_t n = std::size_t(1) << (argc > 1 ? std::stoul(argv[1]) : 24), C = argc > 2 ? std::stoul(argv[2]) : 8, fpe = argc > 3 ? std::stoul(argv[3]) : 0, N = 8192, K = 256;
    const std::string mode = argc > 4 ? argv[4] : "pinned";
    if (const char* g = std::getenv("GR4HIP_BENCH_GUARD_MODE")) gr::hip::options().guard_mode = std::atoi(g); // 0 strict (default), 1 deferred, 2 off
    std::vector<double> taps(K, 1.0 / double(K));
    Graph g;
    hip::register_provider();
    EdgeParameters big;
    big.minBufferSize = std::size_t(1) << 22;
    EdgeParameters pinned = big;
    if (mode != "pageable") pinned.domain = "gpu:hip:0";
    auto& add  = g.emplaceBlock<blocks::math::Add<float>>(property_map{{"n_inputs", std::int64_t(C)}});
    auto& sink = g.emplaceBlock<testing::NullSink<float>>();
    std::vector<DmaSource<std::complex<float>>*> srcs;
    for (std::size_t c = 0; c < C; ++c) {
        auto& src  = g.emplaceBlock<DmaSource<std::complex<float>>>({{"n_samples_max", std::int64_t(n)}});
        auto& fir  = g.emplaceBlock<filter::fir_filter<std::complex<float>>>(property_map{{"b", taps}, {"compute_domain", "gpu:hip:0"s}});
        auto& spec = g.emplaceBlock<blocks::fft::PowerSpectrum<std::complex<float>>>(property_map{{"fftSize", std::int64_t(N)}, {"window", "None"s}, {"compute_domain", "gpu:hip:0"s}});
        if (!g.connect<"out", "in">(src, fir, pinned) || !g.connect<"out", "in">(fir, spec, big) || !g.connect(spec, "out"s, add, "in#"s + std::to_string(c))) return 2;
        srcs.push_back(&src);
    }
    if (!g.connect<"out", "in">(add, sink, pinned)) return 2;
    for (std::size_t c = 0; c < C; ++c) srcs[c]->fill(c);
    hip::Shard shard{0, 1, nullptr, false, fpe};
    const auto runs = hip::plan_sharded(g, shard, 0);
    if (runs.size() != 1) { std::fprintf(stderr, "planner: expected one fan-in run\n"); return 2; }
    auto* run = runs[0];
    scheduler::Simple sched;
    sched.exchange(std::move(g));
    const auto t0 = std::chrono::steady_clock::now();
    if (const auto r = sched.runAndWait(); !r) { std::fprintf(stderr, "%s\n", r.error().message.c_str()); return 3; }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    const double in = double(C) * double(sink._count);
    std::printf("host-fed fan-in (%s, %s edges, %zu frames per exchange): %zu output samples, %zu channels in %.3f s = %.1f Msamples/s aggregate (%.2f GB/s in + %.2f GB/s out over PCIe); %zu launches, %zu of them queued beside the exchange before\n",
                std::string(run->name()).c_str(), mode.c_str(), run->frames_per_exchange(), sink._count, C, dt, in / dt / 1e6, in * 8 / dt / 1e9, double(sink._count) * 4 / dt / 1e9, run->launches(), run->overlapped());
    return sink._count == (n / N) * N ? 0 : 1;
}
