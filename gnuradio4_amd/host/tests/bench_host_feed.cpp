// Host-fed rate of the fused chain through the C++ graph API (developer tool, DESIGN.md "Host feed"):
//   VectorSource<complex<float>> -> fir_filter (gpu) -> PowerSpectrum (gpu) -> NullSink<float>,  planned into one DeviceRun.
// Everything a sample goes through is timed: source loop, host edge FIFO, pinned staging, H2D, the fused kernel, D2H, sink.
//   bench_host_feed [log2_samples = 27] [fftSize = 8192] [ntaps = 256] [mode = dma] [log2_edge_capacity = 24] [grow]
//   (an explicit edge capacity is measured as it is; with "grow" the planner enlarges the run's edges as it does by default: hip::plan's run_edge_items)
// mode: "dma"      page-locked edges at both ends ("hip" provider); the source publishes spans of its output edge without rewriting them, the way a
//                  driver whose DMA engine fills the port buffer does; the copy engine reads and writes the edges in place: no host copy at all
//       "pinned"   page-locked edges, the source memcpy's every sample into its port buffer (VectorSource): one host copy per sample
//       "pageable" ordinary edges: source copy + staging copies through page-locked memory by the copy threads (GR4HIP_COPY_THREADS)
//       "link"     no graph: the same chunk sizes copied host -> device and device -> host on two streams at once, nothing else: what the link gives
#include <chrono>
#include <cstdlib>
#include <cstdio>

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
    void fill() { // once, before the clock starts
        for (std::size_t i = 0, nfill = out.buffer->is_ring() ? out.buffer->capacity : out.buffer->data.size(); i < nfill; ++i) out.buffer->base()[i] = {static_cast<float>(i % 17) - 8.f, static_cast<float>(i % 5)};
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
    const std::size_t n = std::size_t(1) << (argc > 1 ? std::stoul(argv[1]) : 27), N = argc > 2 ? std::stoul(argv[2]) : 8192, K = argc > 3 ? std::stoul(argv[3]) : 256;
    std::vector<double> taps(K, 1.0 / double(K));
    const std::string mode = argc > 4 ? argv[4] : "dma";
    if (const char* g = std::getenv("GR4HIP_BENCH_GUARD_MODE")) gr::hip::options().guard_mode = std::atoi(g); // 0 strict (default), 1 deferred, 2 off
    if (mode == "link") {
        const std::size_t chunk = std::size_t(2) << 20, chunks = n / chunk; // 2 Mi samples: 16 MiB in, 8 MiB out per chunk (what the run moves)
        void *h_in = nullptr, *h_out = nullptr, *d_in = nullptr, *d_out = nullptr;
        gr4hip_stream_t s_in = nullptr, s_out = nullptr;
        if (gr4hip_set_device(0) || gr4hip_malloc_host(&h_in, chunk * 8) || gr4hip_malloc_host(&h_out, chunk * 4) || gr4hip_malloc(&d_in, chunk * 8) || gr4hip_malloc(&d_out, chunk * 4) ||
            gr4hip_stream_create(&s_in) || gr4hip_stream_create(&s_out)) return 2;
        for (int both = 0; both < 2; ++both) {
            const auto t0 = std::chrono::steady_clock::now();
            for (std::size_t c = 0; c < chunks; ++c) {
                gr4hip_memcpy_h2d(d_in, h_in, chunk * 8, s_in);
                if (both) gr4hip_memcpy_d2h(h_out, d_out, chunk * 4, s_out);
            }
            gr4hip_stream_synchronize(s_in);
            gr4hip_stream_synchronize(s_out);
            const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            std::printf("link only, %s: %zu chunks of 16 MiB in %.3f s = %.2f GB/s host -> device%s = %.1f Msamples/s\n", both ? "both directions" : "host -> device alone", chunks, dt,
                        double(chunks * chunk) * 8 / dt / 1e9, both ? " with half of that device -> host beside it" : "", double(chunks * chunk) / dt / 1e6);
        }
        return 0;
    }
    Graph g;
    auto connect_src = [&](auto& src, auto& dst, const EdgeParameters& e) { return static_cast<bool>(g.connect<"out", "in">(src, dst, e)); };
    auto& fir  = g.emplaceBlock<filter::fir_filter<std::complex<float>>>({{"b", taps}, {"compute_domain", "gpu:hip:0"s}});
    auto& spec = g.emplaceBlock<blocks::fft::PowerSpectrum<std::complex<float>>>({{"fftSize", std::int64_t(N)}, {"window", "None"s}, {"compute_domain", "gpu:hip:0"s}});
    auto& sink = g.emplaceBlock<testing::NullSink<float>>();
    EdgeParameters big;
    big.minBufferSize = std::size_t(1) << (argc > 5 ? std::stoul(argv[5]) : 24);
    hip::register_provider();
    EdgeParameters pinned = big;
    if (mode != "pageable") pinned.domain = "gpu:hip:0"; // page-locked edges: the run's copy engine reads / writes them in place
    bool ok = true;
    if (mode == "dma") {
        auto& src = g.emplaceBlock<DmaSource<std::complex<float>>>({{"n_samples_max", std::int64_t(n)}});
        ok = connect_src(src, fir, pinned);
        if (ok) src.fill();
    } else {
        auto& src = g.emplaceBlock<testing::VectorSource<std::complex<float>>>({{"n_samples_max", std::int64_t(n)}});
        src.values.resize(1 << 20);
        for (std::size_t i = 0; i < src.values.size(); ++i) src.values[i] = {static_cast<float>(i % 17) - 8.f, static_cast<float>(i % 5)};
        ok = connect_src(src, fir, pinned);
    }
    if (!ok || !g.connect<"out", "in">(fir, spec, big) || !g.connect<"out", "in">(spec, sink, pinned)) return 2;
    if (argc > 6 && std::string(argv[6]) == "seam") { // no planner: every device block on the per-block seam (synchronous copy-in, kernel, copy-out per work() call)
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        const auto t0 = std::chrono::steady_clock::now();
        if (const auto r = sched.runAndWait(); !r) { std::fprintf(stderr, "%s\n", r.error().message.c_str()); return 3; }
        const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("host-fed, per-block seam (no planner, %s): %zu samples in %.3f s = %.1f Msamples/s\n", mode.c_str(), sink._count, dt, double(sink._count) / dt / 1e6);
        return sink._count == (n / N) * N ? 0 : 1;
    }
    const auto runs = hip::plan(g, 2, (argc > 5 && !(argc > 6 && std::string(argv[6]) == "grow")) ? 0 : std::size_t(1) << 22);
    if (runs.size() != 1) { std::fprintf(stderr, "planner: expected one run\n"); return 2; }
    scheduler::Simple sched;
    sched.exchange(std::move(g));
    const auto t0 = std::chrono::steady_clock::now();
    if (const auto r = sched.runAndWait(); !r) { std::fprintf(stderr, "%s\n", r.error().message.c_str()); return 3; }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("host-fed chain (%s, %s, %zu copy helpers): %zu samples in %.3f s = %.1f Msamples/s (%.2f GB/s in + %.2f GB/s out over PCIe); %zu launches, %zu overlapped, %zu read in place, %zu written in place\n",
                std::string(runs[0]->description()).c_str(), mode.c_str(), hip::CopyPool::instance().helpers(), sink._count, dt, double(sink._count) / dt / 1e6, double(sink._count) * 8 / dt / 1e9,
                double(sink._count) * 4 / dt / 1e9, runs[0]->launches(), runs[0]->overlapped_chunks(), runs[0]->inplace_chunks(), runs[0]->direct_chunks());
    return sink._count == (n / N) * N ? 0 : 1;
}
