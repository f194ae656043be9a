// Host-fed rate of the fused chain through the C++ graph API (developer tool, DESIGN.md "Host feed"):
//   VectorSource<complex<float>> -> fir_filter (gpu) -> PowerSpectrum (gpu) -> NullSink<float>,  planned into one DeviceRun.
// Everything a sample goes through is timed: source loop, host edge FIFO, pinned staging, H2D, the fused kernel, D2H, sink.
//   bench_host_feed [log2_samples = 27] [fftSize = 8192] [ntaps = 256] [pageable]
#include <chrono>
#include <cstdio>

#include <gr4/hip.hpp>

using namespace gr;
using namespace std::string_literals;

int main(int argc, char** argv) {
    const std::size_t n = std::size_t(1) << (argc > 1 ? std::stoul(argv[1]) : 27), N = argc > 2 ? std::stoul(argv[2]) : 8192, K = argc > 3 ? std::stoul(argv[3]) : 256;
    std::vector<double> taps(K, 1.0 / double(K));
    Graph g;
    auto& src = g.emplaceBlock<testing::VectorSource<std::complex<float>>>({{"n_samples_max", std::int64_t(n)}});
    src.values.resize(1 << 20);
    for (std::size_t i = 0; i < src.values.size(); ++i) src.values[i] = {static_cast<float>(i % 17) - 8.f, static_cast<float>(i % 5)};
    auto& fir  = g.emplaceBlock<filter::fir_filter<std::complex<float>>>({{"b", taps}, {"compute_domain", "gpu:hip:0"s}});
    auto& spec = g.emplaceBlock<blocks::fft::PowerSpectrum<std::complex<float>>>({{"fftSize", std::int64_t(N)}, {"window", "None"s}, {"compute_domain", "gpu:hip:0"s}});
    auto& sink = g.emplaceBlock<testing::NullSink<float>>();
    EdgeParameters big;
    big.minBufferSize = std::size_t(1) << 22;
    hip::register_provider();
    EdgeParameters pinned = big;
    if (!(argc > 4 && std::string(argv[4]) == "pageable")) pinned.domain = "gpu:hip:0"; // page-locked input edge: the run's copy engine reads it in place
    if (!g.connect<"out", "in">(src, fir, pinned) || !g.connect<"out", "in">(fir, spec, big) || !g.connect<"out", "in">(spec, sink, big)) return 2;
    const auto runs = hip::plan(g);
    if (runs.size() != 1) { std::fprintf(stderr, "planner: expected one run\n"); return 2; }
    scheduler::Simple sched;
    sched.exchange(std::move(g));
    const auto t0 = std::chrono::steady_clock::now();
    if (const auto r = sched.runAndWait(); !r) { std::fprintf(stderr, "%s\n", r.error().message.c_str()); return 3; }
    const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("host-fed chain (%s): %zu samples in %.3f s = %.1f Msamples/s (%.2f GB/s in + %.2f GB/s out over PCIe); %zu launches, %zu overlapped, %zu read in place\n",
                std::string(runs[0]->description()).c_str(), sink._count, dt, double(sink._count) / dt / 1e6, double(sink._count) * 8 / dt / 1e9, double(sink._count) * 4 / dt / 1e9,
                runs[0]->launches(), runs[0]->overlapped_chunks(), runs[0]->inplace_chunks());
    return sink._count == (n / N) * N ? 0 : 1;
}
