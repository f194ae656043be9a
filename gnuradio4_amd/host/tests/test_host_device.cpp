// Device-side test of the host layer: the compute_domain = "gpu:hip" seam and the fusing HIP-stream run.
//   test_host_device <in_c32.bin> <taps.bin> <fftSize> <out_prefix>
// in_c32.bin: interleaved complex<float> stream; taps.bin: float taps.  Writes <out_prefix>_{fir,chain,chain_unfused,math}.bin.
// Exit code 0: all graphs ran; 3: a device block reported work::Status::ERROR (what must happen on a box without a GPU).
#include <cstdio>
#include <fstream>
#include <iostream>

#include <gr4/hip.hpp>

using namespace gr;
using namespace std::string_literals;

template <typename T>
std::vector<T> load(const char* path) {
    std::ifstream f(path, std::ios::binary | std::ios::ate);
    if (!f) { std::fprintf(stderr, "cannot open %s\n", path); std::exit(2); }
    const auto     bytes = static_cast<std::size_t>(f.tellg());
    std::vector<T> v(bytes / sizeof(T));
    f.seekg(0);
    f.read(reinterpret_cast<char*>(v.data()), static_cast<std::streamsize>(v.size() * sizeof(T)));
    return v;
}
template <typename T>
void dump(const std::string& path, const std::vector<T>& v) {
    std::ofstream f(path, std::ios::binary);
    f.write(reinterpret_cast<const char*>(v.data()), static_cast<std::streamsize>(v.size() * sizeof(T)));
}

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s in_c32.bin taps.bin fftSize out_prefix\n", argv[0]); return 2; }
    const auto        x    = load<std::complex<float>>(argv[1]);
    const auto        taps = load<float>(argv[2]);
    const std::size_t N    = std::stoul(argv[3]);
    const std::string out  = argv[4];
    const std::vector<double> tapsd(taps.begin(), taps.end());
    int errors = 0;

    { // 1. one device block inside a host graph: the seam in Block::dispatchProcessing offloads fir_filter<complex<float>>
        Graph g;
        auto& src = g.emplaceBlock<testing::VectorSource<std::complex<float>>>();
        src.values = x;
        auto& fir  = g.emplaceBlock<filter::fir_filter<std::complex<float>>>({{"b", tapsd}, {"compute_domain", "gpu:hip:0"s}});
        fir._log   = [](std::string_view m) { std::cerr << "[log] " << m << "\n"; };
        auto& sink = g.emplaceBlock<testing::VectorSink<std::complex<float>>>();
        if (!g.connect<"out", "in">(src, fir) || !g.connect<"out", "in">(fir, sink)) return 2;
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        if (const auto r = sched.runAndWait(); !r) { std::cerr << "fir graph: " << r.error().message << "\n"; ++errors; }
        else dump(out + "_fir.bin", sink._samples);
        hip::release(fir);
    }
    if (errors) return 3; // no device: fail loudly, never a host fallback

    { // 2. integer math block on the device: bit-exact
        Graph g;
        auto& src = g.emplaceBlock<testing::VectorSource<std::int32_t>>({{"n_samples_max", std::int64_t(100000)}});
        src.values = {2147483647, -5, 7, 123456789};
        auto& mul  = g.emplaceBlock<blocks::math::MultiplyConst<std::int32_t>>({{"value", std::int64_t(3)}, {"compute_domain", "gpu:hip"s}});
        auto& sink = g.emplaceBlock<testing::VectorSink<std::int32_t>>();
        g.connect<"out", "in">(src, mul);
        g.connect<"out", "in">(mul, sink);
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        if (!sched.runAndWait()) ++errors;
        dump(out + "_math.bin", sink._samples);
        hip::release(mul);
    }

    for (int fused = 1; fused >= 0; --fused) { // 3. fir -> PowerSpectrum as a device run: fused into one launch, or stage by stage
        Graph g;
        auto& src = g.emplaceBlock<testing::VectorSource<std::complex<float>>>();
        src.values = x;
        auto& fir  = g.emplaceBlock<filter::fir_filter<std::complex<float>>>({{"b", tapsd}, {"compute_domain", "gpu:hip:0"s}});
        auto& spec = g.emplaceBlock<blocks::fft::PowerSpectrum<std::complex<float>>>({{"fftSize", std::int64_t(N)}, {"window", fused ? "None"s : "Hann"s}, {"compute_domain", "gpu:hip:0"s}});
        auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
        g.connect<"out", "in">(src, fir);
        g.connect<"out", "in">(fir, spec);
        g.connect<"out", "in">(spec, sink);
        auto& run = hip::fuse_chain(g, fir, spec); // both blocks leave the schedule; one DeviceRun takes their place
        std::printf("device run: %s  (%zu stage%s)\n", std::string(run.description()).c_str(), run.stages().size(), run.stages().size() == 1 ? "" : "s");
        if (run.stages().size() != 1 || run.description() != "chain_fir_fft_mag2") ++errors;
        const int algo = static_cast<const hip::ChainStage*>(run.stages()[0].get())->algo();
        std::printf("chain algo in use: %d\n", algo);
        if (fused && N == 8192 && taps.size() <= 256 && algo != GR4HIP_CHAIN_FUSED_FD) ++errors; // headline shape must take the fused kernel
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        if (const auto r = sched.runAndWait(); !r) { std::cerr << "chain graph: " << r.error().message << "\n"; ++errors; }
        if (sink._samples.size() != (x.size() / N) * N) ++errors;
        dump(out + (fused ? "_chain.bin" : "_chain_hann.bin"), sink._samples);
    }
    { // 3b. the planner finds the device chains by itself: (a) fir -> PowerSpectrum collapses into the fused kernel, (b) MultiplyConst -> fir_filter<float>
      //     becomes a two-stage run on one stream, (c) a host block between two device blocks splits the chain
        Graph g;
        auto& src  = g.emplaceBlock<testing::VectorSource<std::complex<float>>>();
        src.values = x;
        auto& fir  = g.emplaceBlock<filter::fir_filter<std::complex<float>>>({{"b", tapsd}, {"compute_domain", "gpu:hip:0"s}});
        auto& spec = g.emplaceBlock<blocks::fft::PowerSpectrum<std::complex<float>>>({{"fftSize", std::int64_t(N)}, {"window", "BlackmanHarris"s}, {"compute_domain", "gpu:hip:0"s}});
        auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
        g.connect<"out", "in">(src, fir);
        g.connect<"out", "in">(fir, spec);
        g.connect<"out", "in">(spec, sink);
        auto& fsrc  = g.emplaceBlock<testing::VectorSource<float>>({{"n_samples_max", std::int64_t(200000)}});
        fsrc.values = {1.f, -2.f, 3.f, 0.5f, 0.25f};
        auto& mul   = g.emplaceBlock<blocks::math::MultiplyConst<float>>({{"value", 2.0}, {"compute_domain", "gpu:hip:0"s}});
        auto& ffir  = g.emplaceBlock<filter::fir_filter<float>>({{"b", std::vector<double>{0.5, 0.25, 0.25}}, {"compute_domain", "gpu:hip:0"s}});
        auto& host  = g.emplaceBlock<blocks::math::AddConst<float>>({{"value", 1.0}}); // stays on the host
        auto& mul2  = g.emplaceBlock<blocks::math::MultiplyConst<float>>({{"value", 3.0}, {"compute_domain", "gpu:hip:0"s}});
        auto& fsink = g.emplaceBlock<testing::VectorSink<float>>();
        g.connect<"out", "in">(fsrc, mul);
        g.connect<"out", "in">(mul, ffir);
        g.connect<"out", "in">(ffir, host);
        g.connect<"out", "in">(host, mul2);
        g.connect<"out", "in">(mul2, fsink);
        const auto runs = hip::plan(g);
        std::printf("planner: %zu runs:", runs.size());
        for (auto* r : runs) std::printf(" [%s]", std::string(r->description()).c_str());
        std::printf("\n");
        if (runs.size() != 2 || runs[0]->description() != "chain_fir_fft_mag2" || runs[1]->description() != "math_const -> fir_f32") ++errors;
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        if (const auto r = sched.runAndWait(); !r) { std::cerr << "planned graph: " << r.error().message << "\n"; ++errors; }
        dump(out + "_chain_planned_bh.bin", sink._samples);
        dump(out + "_planned_float.bin", fsink._samples);
        hip::release(mul2);
    }

    { // 4. the GPU-resident BufferLike ring: spans that wrap the physical end stay contiguous; two readers, back-pressure
        hip::CircularBuffer<float> ring(1 << 16);
        auto                       w = ring.new_writer();
        auto                       r1 = ring.new_reader(), r2 = ring.new_reader();
        const std::size_t          cap = ring.size(), chunk = cap / 2 + 4096; // the second chunk wraps
        std::vector<float>         host(chunk), back(chunk);
        std::size_t                seq = 0;
        for (int round = 0; round < 5 && !errors; ++round) {
            for (auto& v : host) v = static_cast<float>(seq++);
            auto span = w.reserve(chunk);
            hip::check(gr4hip_memcpy_h2d(span.data(), host.data(), chunk * sizeof(float), nullptr), "ring h2d");
            hip::check(gr4hip_stream_synchronize(nullptr), "sync");
            w.publish(chunk);
            if (!w.tryReserve(chunk).empty()) ++errors; // both readers still hold the chunk: no room for another one
            for (auto* r : {&r1, &r2}) {
                auto in = r->get(chunk);
                if (in.size() != chunk) { ++errors; break; }
                hip::check(gr4hip_memcpy_d2h(back.data(), in.data(), chunk * sizeof(float), nullptr), "ring d2h"); // ONE copy, also when wrapping
                hip::check(gr4hip_stream_synchronize(nullptr), "sync");
                if (back != host) ++errors;
                if (!r->consume(chunk)) ++errors;
            }
        }
        std::printf("device ring: capacity %zu floats, %s\n", cap, errors ? "FAILED" : "wrapping spans contiguous");
    }
    std::printf(errors ? "host-device: %d FAILURES\n" : "host-device: all graphs ran\n", errors);
    return errors ? 1 : 0;
}
