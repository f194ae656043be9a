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

// one block between a VectorSource and a VectorSink, on the host path or behind the compute_domain seam
template <typename BlockT, typename TIn, typename TOut>
std::vector<TOut> run_one(property_map settings, const std::vector<TIn>& input, bool device, int& errors) {
    Graph g;
    auto& src  = g.emplaceBlock<testing::VectorSource<TIn>>();
    src.values = input;
    if (device) settings["compute_domain"] = "gpu:hip:0"s;
    auto& blk  = g.emplaceBlock<BlockT>(settings);
    blk._log   = [](std::string_view m) { std::cerr << "[log] " << m << "\n"; };
    auto& sink = g.emplaceBlock<testing::VectorSink<TOut>>();
    if (!g.connect<"out", "in">(src, blk) || !g.connect<"out", "in">(blk, sink)) { ++errors; return {}; }
    scheduler::Simple sched;
    sched.exchange(std::move(g));
    if (const auto r = sched.runAndWait(); !r) { std::cerr << "run_one: " << r.error().message << "\n"; ++errors; }
    if (device && !blk._device_state) ++errors; // the seam must have been taken (the state goes with the block)
    return sink._samples;
}
template <typename T>
double max_rel(const std::vector<T>& got, const std::vector<T>& want) { // max |got - want| / max(|want|, rms(want))
    if (got.size() != want.size() || want.empty()) return 1e30;
    double rms = 0;
    for (const auto& v : want) rms += std::norm(std::complex<double>(v));
    rms = std::sqrt(rms / double(want.size()));
    double worst = 0;
    for (std::size_t i = 0; i < want.size(); ++i) worst = std::max(worst, std::abs(std::complex<double>(got[i]) - std::complex<double>(want[i])) / std::max(std::abs(std::complex<double>(want[i])), rms));
    return worst;
}

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s in_c32.bin taps.bin fftSize out_prefix\n", argv[0]); return 2; }
    const auto        x    = load<std::complex<float>>(argv[1]);
    const auto        taps = load<float>(argv[2]);
    const std::size_t N    = std::stoul(argv[3]);
    const std::string out  = argv[4];
    const std::vector<double> tapsd(taps.begin(), taps.end());
    int errors = 0;

    { // 0. port domains: a CPU port cannot be wired to a GPU port, the error names the converter blocks (no device needed: nothing is allocated)
        Graph g;
        auto& src = g.emplaceBlock<testing::VectorSource<float>>();
        auto& dev = g.emplaceBlock<hip::OnDevice<filter::fir_filter<float>>>({{"b", std::vector<double>{1.0}}});
        auto& d2h = g.emplaceBlock<hip::D2H<float>>();
        const auto r = g.connect<"out", "in">(src, dev);
        const bool ok = !r.has_value() && r.error().message.find("different computing domains (CPU -> GPU)") != std::string::npos && r.error().message.find("gr::hip::H2D") != std::string::npos;
        const auto r2 = g.connect<"out", "in">(d2h, dev); // D2H.out is a CPU port
        std::printf("port domains: %s\n", ok && !r2.has_value() ? "CPU -> GPU refused, converter required" : "FAILED");
        if (!ok || r2.has_value()) ++errors;
    }

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
      //     becomes ONE stage (the gain in the filter's launch), (c) a host block between two device blocks splits the chain
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
        if (runs.size() != 2 || runs[0]->description() != "chain_fir_fft_mag2" || runs[1]->description() != "fir_f32[pre: mul]") ++errors; // (the gain rides in the filter's launch)
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        if (const auto r = sched.runAndWait(); !r) { std::cerr << "planned graph: " << r.error().message << "\n"; ++errors; }
        dump(out + "_chain_planned_bh.bin", sink._samples);
        dump(out + "_planned_float.bin", fsink._samples);
        hip::release(mul2);
    }
    { // 3b''. the chain between page-locked edges ("hip" provider) at both ends: the run's copy engine reads the input edge and writes the output edge in
      //       place, several chunks in flight, the edges small enough (4 frames) that their storage fills and compacts many times on the way.
      //       Sample for sample the result of the same graph with ordinary edges (staged through the run's own page-locked buffers).
        hip::register_provider();
        std::size_t launches = 0, inplace = 0, direct = 0, overlapped = 0;
        const auto run_chain = [&](bool locked_edges) {
            Graph g;
            auto& src  = g.emplaceBlock<testing::VectorSource<std::complex<float>>>({{"n_samples_max", std::int64_t(96 * N + 77)}});
            src.values = x; // repeated cyclically
            auto& fir  = g.emplaceBlock<filter::fir_filter<std::complex<float>>>({{"b", tapsd}, {"compute_domain", "gpu:hip:0"s}});
            auto& spec = g.emplaceBlock<blocks::fft::PowerSpectrum<std::complex<float>>>({{"fftSize", std::int64_t(N)}, {"window", "BlackmanHarris"s}, {"compute_domain", "gpu:hip:0"s}});
            auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
            EdgeParameters e;
            e.minBufferSize = 4 * N;
            if (locked_edges) e.domain = "gpu:hip:0";
            if (!g.connect<"out", "in">(src, fir, e) || !g.connect<"out", "in">(fir, spec) || !g.connect<"out", "in">(spec, sink, e)) ++errors;
            const auto runs = hip::plan(g, 2, 0); // (0: the planner leaves these deliberately small edges alone)
            scheduler::Simple sched;
            sched.exchange(std::move(g));
            if (const auto r = sched.runAndWait(); !r) { std::cerr << "page-locked graph: " << r.error().message << "\n"; ++errors; }
            if (runs.size() != 1) { ++errors; return std::vector<float>{}; }
            if (locked_edges) { launches = runs[0]->launches(); inplace = runs[0]->inplace_chunks(); direct = runs[0]->direct_chunks(); overlapped = runs[0]->overlapped_chunks(); }
            else if (runs[0]->inplace_chunks() || runs[0]->direct_chunks()) ++errors;
            return sink._samples;
        };
        const auto staged = run_chain(false), in_place = run_chain(true);
        const bool same   = staged.size() == 96 * N && in_place == staged;
        std::printf("page-locked edges: %zu launches, %zu chunks read in place, %zu written in place, %zu overlapped, output %s the staged run\n", launches, inplace, direct, overlapped,
                    same ? "equals" : "DIFFERS from");
        if (!same || launches < 1 || inplace < launches || direct < launches) ++errors; // (round 6: edges this small are gathered in the device ring -- several pieces read and written in place per launch)
    }
    { // 3b'. a tee'd device edge: fir.out feeds the spectrum block AND a host sink -> the planner must not fuse across it (the filtered samples have to
      //      reach the host edge); both readers see every sample
        Graph g;
        auto& src  = g.emplaceBlock<testing::VectorSource<std::complex<float>>>();
        src.values = x;
        auto& fir  = g.emplaceBlock<filter::fir_filter<std::complex<float>>>({{"b", tapsd}, {"compute_domain", "gpu:hip:0"s}});
        auto& spec = g.emplaceBlock<blocks::fft::PowerSpectrum<std::complex<float>>>({{"fftSize", std::int64_t(N)}, {"window", "Hann"s}, {"compute_domain", "gpu:hip:0"s}});
        auto& s1   = g.emplaceBlock<testing::VectorSink<float>>();
        auto& s2   = g.emplaceBlock<testing::VectorSink<std::complex<float>>>();
        g.connect<"out", "in">(src, fir);
        g.connect<"out", "in">(fir, spec);
        g.connect<"out", "in">(fir, s2); // second reader of fir.out
        g.connect<"out", "in">(spec, s1);
        const auto runs = hip::plan(g);
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        if (const auto r = sched.runAndWait(); !r) { std::cerr << "tee graph: " << r.error().message << "\n"; ++errors; }
        std::printf("tee'd device edge: %zu fused runs, %zu spectra samples, %zu filtered samples\n", runs.size(), s1._samples.size(), s2._samples.size());
        if (!runs.empty() || s1._samples.size() != (x.size() / N) * N || s2._samples.size() != x.size()) ++errors;
        dump(out + "_tee_fir.bin", s2._samples);
        hip::release(fir);
        hip::release(spec);
    }
    { // 3c. GPU-domain ports: src -> H2D -> fir (GPU ports) -> PowerSpectrum (GPU ports) -> D2H -> sink.  The edges between the converters are
      //     rings in HBM; the edge INTO H2D is allocated from the "hip" provider (pinned pages), so the converter copies without staging
        hip::register_provider();
        Graph g;
        auto& src  = g.emplaceBlock<testing::VectorSource<std::complex<float>>>();
        src.values = x;
        auto& h2d  = g.emplaceBlock<hip::H2D<std::complex<float>>>();
        auto& fir  = g.emplaceBlock<hip::OnDevice<filter::fir_filter<std::complex<float>>>>({{"b", tapsd}, {"name", "fir@gpu"s}});
        auto& spec = g.emplaceBlock<hip::OnDevice<blocks::fft::PowerSpectrum<std::complex<float>>>>({{"fftSize", std::int64_t(N)}, {"window", "Hann"s}});
        auto& d2h  = g.emplaceBlock<hip::D2H<float>>();
        auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
        bool  wired = g.connect<"out", "in">(src, h2d, EdgeParameters{.domain = "gpu:hip:0"}).has_value();
        wired       = wired && g.connect<"out", "in">(h2d, fir).has_value() && g.connect<"out", "in">(fir, spec).has_value() && g.connect<"out", "in">(spec, d2h).has_value() &&
                g.connect<"out", "in">(d2h, sink).has_value();
        if (!wired) ++errors;
        const bool pinned_edge = h2d.in.buffer && h2d.in.buffer->resource() == hip::pinned_resource();
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        if (const auto r = sched.runAndWait(); !r) { std::cerr << "gpu-port graph: " << r.error().message << "\n"; ++errors; }
        std::printf("gpu ports: %zu launches fir, %zu launches spectrum, H2D moved %zu bytes (%zu staged), input edge %s\n", fir._launches, spec._launches, h2d._bytes, h2d._staged_bytes,
                    pinned_edge ? "pinned by the hip provider" : "pageable");
        if (!pinned_edge || h2d._staged_bytes != 0 || h2d._bytes != (x.size() / N) * N * sizeof(std::complex<float>) + (x.size() % N) * sizeof(std::complex<float>)) ++errors;
        if (sink._samples.size() != (x.size() / N) * N) ++errors;
        dump(out + "_gpu_ports_hann.bin", sink._samples);
    }

    { // 3d. a tee on a GPU-DOMAIN edge (round 5): fir.out (GPU port) feeds the spectrum block (GPU port) AND a D2H converter -> both readers hold their own cursor on the
      //     SAME ring in HBM (one writer -> N readers, CircularBuffer.hpp:880-946): no copy of the filtered stream, the writer waits for the slower of the two
        Graph g;
        auto& src  = g.emplaceBlock<testing::VectorSource<std::complex<float>>>();
        src.values = x;
        auto& h2d  = g.emplaceBlock<hip::H2D<std::complex<float>>>();
        auto& fir  = g.emplaceBlock<hip::OnDevice<filter::fir_filter<std::complex<float>>>>({{"b", tapsd}, {"name", "fir@gpu"s}});
        auto& spec = g.emplaceBlock<hip::OnDevice<blocks::fft::PowerSpectrum<std::complex<float>>>>({{"fftSize", std::int64_t(N)}, {"window", "None"s}});
        auto& d2hs = g.emplaceBlock<hip::D2H<float>>();
        auto& d2hf = g.emplaceBlock<hip::D2H<std::complex<float>>>();
        auto& s1   = g.emplaceBlock<testing::VectorSink<float>>();
        auto& s2   = g.emplaceBlock<testing::VectorSink<std::complex<float>>>();
        bool  wired = g.connect<"out", "in">(src, h2d).has_value() && g.connect<"out", "in">(h2d, fir).has_value() && g.connect<"out", "in">(fir, spec).has_value() &&
                     g.connect<"out", "in">(fir, d2hf).has_value() /* the second reader of fir.out */ && g.connect<"out", "in">(spec, d2hs).has_value() &&
                     g.connect<"out", "in">(d2hs, s1).has_value() && g.connect<"out", "in">(d2hf, s2).has_value();
        if (!wired) ++errors;
        const bool   same_ring = wired && fir.out.buffer && fir.out.buffer->views.size() == 1 && d2hf.in.buffer && d2hf.in.buffer->is_view &&
                               d2hf.in.buffer->read_span(0).data() == fir.out.buffer->read_span(0).data();
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        if (const auto r = sched.runAndWait(); !r) { std::cerr << "gpu-domain tee: " << r.error().message << "\n"; ++errors; }
        std::printf("gpu-domain tee: %s, %zu spectra samples, %zu filtered samples\n", same_ring ? "2 readers on one ring in HBM" : "NOT one ring", s1._samples.size(), s2._samples.size());
        if (!same_ring || s1._samples.size() != (x.size() / N) * N || s2._samples.size() != x.size()) ++errors;
        dump(out + "_gpu_tee_fir.bin", s2._samples);
        dump(out + "_gpu_tee_spec.bin", s1._samples);
    }

    { // 3e. the page-locked host edge as a RING (round 5; north star: "pinned hipMemcpyAsync into a GPU-resident double-mapped CircularBuffer"): a CPU-domain edge of the
      //     "hip" provider at the reference's default 65536 items is `capacity` items mapped twice (gr4hip_host_ring_create); 20 x its capacity streams through
      //     src -> [ring edge] -> H2D -> [HBM ring] -> D2H -> [ring edge] -> sink: both host edges wrap many times, nothing is moved to their front, every sample arrives
        hip::register_provider();
        Graph g;
        std::vector<float> ramp(20 * 65536 + 123);
        for (std::size_t i = 0; i < ramp.size(); ++i) ramp[i] = static_cast<float>(static_cast<std::int32_t>((i * 2654435761u) >> 8) % 100003);
        auto& src  = g.emplaceBlock<testing::VectorSource<float>>();
        src.values = ramp;
        auto& h2d  = g.emplaceBlock<hip::H2D<float>>();
        auto& d2h  = g.emplaceBlock<hip::D2H<float>>();
        auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
        bool  wired = g.connect<"out", "in">(src, h2d, EdgeParameters{.domain = "gpu:hip:0"}).has_value() && g.connect<"out", "in">(h2d, d2h).has_value() &&
                     g.connect<"out", "in">(d2h, sink, EdgeParameters{.domain = "gpu:hip:0"}).has_value();
        if (!wired) ++errors;
        const bool rings = wired && h2d.in.buffer->is_ring() && d2h.out.buffer->is_ring() && h2d.in.buffer->capacity == 65536 &&
                           h2d.in.buffer->base()[5] == h2d.in.buffer->base()[5 + 65536]; // (the second mapping IS the first)
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        if (const auto r = sched.runAndWait(); !r) { std::cerr << "host ring edge: " << r.error().message << "\n"; ++errors; }
        const bool same = sink._samples == ramp;
        std::printf("host ring edge: %s, %zu samples through 65536-item edges (%zu wraps), %s, %zu bytes staged by H2D\n", rings ? "page-locked double-mapped rings" : "NOT rings", ramp.size(),
                    ramp.size() / 65536, same ? "every sample arrived" : "DATA MISMATCH", h2d._staged_bytes);
        if (!rings || !same || h2d._staged_bytes != 0) ++errors;
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
    { // 5. every other hot-path block behind the seam, against the host body of the same block definition (the oracle comparison of the kernels
      //    themselves is tests/test_gpu_parity.py; this checks the wiring: settings -> handle, chunking, state carried across work() calls)
        std::vector<float> xf(300000);
        std::uint32_t      lcg = 12345u;
        for (auto& v : xf) { lcg = lcg * 1664525u + 1013904223u; v = static_cast<float>(static_cast<std::int32_t>(lcg >> 8) % 2001 - 1000) / 1000.f; }
        std::vector<std::int32_t> xi(xf.size());
        for (std::size_t i = 0; i < xi.size(); ++i) xi[i] = static_cast<std::int32_t>(xf[i] * 2.0e9f);
        const auto report = [&](const char* what, double err, double tol) {
            std::printf("seam %-34s max rel err %.3g%s\n", what, err, err <= tol ? "" : "  FAILED");
            if (!(err <= tol)) ++errors;
        };
        const std::vector<double> bq_b{0.020083365564211, 0.040166731128423, 0.020083365564211}, bq_a{1.0, -1.561018075800718, 0.641351538057563}; // qa_filter.cpp:86-87
        {
            using B = filter::iir_filter<float, filter::IIRForm::DF_II>;
            const property_map cfg{{"b", bq_b}, {"a", bq_a}};
            report("iir_filter<float, DF_II>", max_rel(run_one<B, float, float>(cfg, xf, true, errors), run_one<B, float, float>(cfg, xf, false, errors)), 1e-5);
            using BT = filter::iir_filter<float, filter::IIRForm::DF_I_TRANSPOSED>;
            report("iir_filter<float, DF_I_TRANSPOSED>", max_rel(run_one<BT, float, float>(cfg, xf, true, errors), run_one<BT, float, float>(cfg, xf, false, errors)), 1e-5);
        }
        {
            using B = filter::Decimator<std::int32_t>;
            const property_map cfg{{"decim", std::int64_t(7)}};
            const auto d = run_one<B, std::int32_t, std::int32_t>(cfg, xi, true, errors), h = run_one<B, std::int32_t, std::int32_t>(cfg, xi, false, errors);
            report("Decimator<int32> decim 7", d == h && d.size() == xi.size() / 7 ? 0.0 : 1.0, 0.0);
        }
        for (const std::size_t L : {std::size_t(2), std::size_t(3), std::size_t(8), std::size_t(7)}) { // interpolating FIR (north_star; Resampling<1, L>): device polyphase kernel vs the host body
            using B = filter::fir_interpolator<float>;
            std::vector<float> bt(91);
            for (std::size_t k = 0; k < bt.size(); ++k) bt[k] = static_cast<float>((0.54 - 0.46 * std::cos(2 * std::numbers::pi * double(k) / 90.0)) / 49.0);
            const property_map cfg{{"b", bt}, {"interpolate", std::int64_t(L)}};
            std::vector<float> xs(xf.begin(), xf.begin() + 20000);
            const auto d = run_one<B, float, float>(cfg, xs, true, errors), h = run_one<B, float, float>(cfg, xs, false, errors);
            report(("fir_interpolator<float> x" + std::to_string(L)).c_str(), d.size() == xs.size() * L && h.size() == d.size() ? max_rel(d, h) : 1e30, 1e-5);
        }
        {
            using B = blocks::math::Rotator<std::complex<float>>;
            std::vector<std::complex<float>> xc(100000);
            for (std::size_t i = 0; i < xc.size(); ++i) xc[i] = {xf[2 * i], xf[2 * i + 1]};
            const property_map cfg{{"phase_increment", 0.1}};
            // the device block evaluates the phase in closed form (float64: carried + (i + 1) inc), the host body is the reference's float accumulator, which drifts
            // ~1e-7 rad per step: compare the device with the float64 phase over the whole stream and with the host body while its drift is still below the bar
            const auto dv = run_one<B, std::complex<float>, std::complex<float>>(cfg, xc, true, errors), hv = run_one<B, std::complex<float>, std::complex<float>>(cfg, xc, false, errors);
            std::vector<std::complex<float>> want(xc.size());
            const double inc = static_cast<double>(0.1f);
            for (std::size_t i = 0; i < xc.size(); ++i) want[i] = static_cast<std::complex<float>>(std::complex<double>(xc[i]) * std::polar(1.0, std::fmod(static_cast<double>(i + 1) * inc, 2.0 * std::numbers::pi)));
            report("Rotator<complex<float>>", dv.size() == want.size() ? max_rel(dv, want) : 1e30, 1e-5);
            // (with inc = 0.1f the float accumulator is already 6e-5 rad off after 1000 steps: same rounding direction inside a binade)
            report("Rotator<complex<float>> vs host body (first 64)", max_rel(std::vector<std::complex<float>>(dv.begin(), dv.begin() + 64), std::vector<std::complex<float>>(hv.begin(), hv.begin() + 64)), 1e-5);
        }
        { // the float64 instantiations the reference registers (fir_filter / iir_filter<double>, Rotator<complex<double>>, FFT<double>): FP64 kernels behind the same seam;
          // the device differs from the host body of the same block only by the order of its float64 sums
            std::vector<double> xd(xf.begin(), xf.begin() + 100000);
            for (std::size_t i = 0; i < xd.size(); ++i) xd[i] += 1e-9 * static_cast<double>(i % 977); // not float-representable
            std::vector<double> bt(77);
            for (std::size_t k = 0; k < bt.size(); ++k) bt[k] = (0.54 - 0.46 * std::cos(2 * std::numbers::pi * double(k) / 76.0)) / 41.0;
            using F = filter::fir_filter<double>;
            report("fir_filter<double> 77 taps", max_rel(run_one<F, double, double>({{"b", bt}}, xd, true, errors), run_one<F, double, double>({{"b", bt}}, xd, false, errors)), 1e-12);
            using I = filter::iir_filter<double, filter::IIRForm::DF_I>;
            const property_map icfg{{"b", bq_b}, {"a", bq_a}};
            report("iir_filter<double, DF_I>", max_rel(run_one<I, double, double>(icfg, xd, true, errors), run_one<I, double, double>(icfg, xd, false, errors)), 1e-11);
            using R = blocks::math::Rotator<std::complex<double>>;
            std::vector<std::complex<double>> xc(50000);
            for (std::size_t i = 0; i < xc.size(); ++i) xc[i] = {xd[2 * i], xd[2 * i + 1]};
            const property_map rcfg{{"phase_increment", 0.1}};
            report("Rotator<complex<double>>", max_rel(run_one<R, std::complex<double>, std::complex<double>>(rcfg, xc, true, errors), run_one<R, std::complex<double>, std::complex<double>>(rcfg, xc, false, errors)), 1e-10);
            using FF = blocks::fft::FFT<double, DataSet<double>>;
            const property_map fcfg{{"fftSize", std::int64_t(1024)}, {"window", "Hann"s}};
            const auto dd = run_one<FF, double, DataSet<double>>(fcfg, xd, true, errors), dh = run_one<FF, double, DataSet<double>>(fcfg, xd, false, errors);
            double worst = dd.size() == xd.size() / 1024 && dh.size() == dd.size() ? 0.0 : 1e30;
            for (std::size_t f = 0; f < dd.size() && worst < 1e29; ++f)
                for (std::size_t i : {std::size_t(0), std::size_t(2), std::size_t(3)}) { // magnitude, Re, Im (phase is noise in empty bins; pinned bin by bin in tests/test_gpu_parity.py)
                    const auto a = dd[f].signalValues(i), b = dh[f].signalValues(i);
                    worst = std::max(worst, max_rel(std::vector<double>(a.begin(), a.end()), std::vector<double>(b.begin(), b.end())));
                }
            report("FFT<double> 1024 Hann -> DataSet<double>", worst, 1e-11);
        }
        for (const char* kind : {"FIR", "IIR"}) {
            const property_map cfg{{"filter_type", std::string(kind)}, {"filter_response", "LOWPASS"s}, {"filter_order", std::int64_t(4)}, {"f_low", 100.0}, {"sample_rate", 1000.0},
                                   {"iir_design_method", "CHEBYSHEV1"s}, {"fir_design_method", "Hamming"s}, {"decimate", std::int64_t(5)}};
            using B = filter::BasicDecimatingFilter<float>;
            const auto d = run_one<B, float, float>(cfg, xf, true, errors), h = run_one<B, float, float>(cfg, xf, false, errors);
            report((std::string("BasicDecimatingFilter<float> ") + kind + " /5").c_str(), d.size() == xf.size() / 5 ? max_rel(d, h) : 1e30, 1e-5);
            dump(out + "_basic_" + (std::string(kind) == "FIR" ? "fir" : "iir") + "5.bin", d);
            auto cfg1 = cfg;
            cfg1.erase("decimate");
            using B1 = filter::BasicFilter<float>;
            report((std::string("BasicFilter<float> ") + kind).c_str(), max_rel(run_one<B1, float, float>(cfg1, xf, true, errors), run_one<B1, float, float>(cfg1, xf, false, errors)), 1e-5);
        }
        dump(out + "_basic_in.bin", xf);
        { // N inputs -> 1 output: Add<int32> wraps like the C++ sum, Multiply<float>
            for (int dev = 1; dev >= 0; --dev) {
                static std::vector<std::int32_t> keep[2];
                Graph g;
                auto& add = g.emplaceBlock<blocks::math::Add<std::int32_t>>(dev ? property_map{{"n_inputs", std::int64_t(3)}, {"compute_domain", "gpu:hip:0"s}} : property_map{{"n_inputs", std::int64_t(3)}});
                for (std::size_t i = 0; i < 3; ++i) {
                    auto& src  = g.emplaceBlock<testing::VectorSource<std::int32_t>>();
                    src.values.assign(xi.begin() + static_cast<std::ptrdiff_t>(i * 1000), xi.begin() + static_cast<std::ptrdiff_t>(i * 1000 + 150000));
                    if (!g.connect(src, "out"s, add, "in#"s + std::to_string(i))) ++errors;
                }
                auto& sink = g.emplaceBlock<testing::VectorSink<std::int32_t>>();
                g.connect<"out", "in">(add, sink);
                scheduler::Simple sched;
                sched.exchange(std::move(g));
                if (!sched.runAndWait()) ++errors;
                if (dev && !add._device_state) ++errors;
                if (dev) hip::release(add);
                keep[dev] = sink._samples;
                if (!dev) report("Add<int32> n_inputs = 3", keep[0] == keep[1] && keep[0].size() == 150000 ? 0.0 : 1.0, 0.0);
            }
        }
        { // gr::UncertainValue<float | double> samples ({value, uncertainty}; Math.hpp:25-28, 68-71): DivideConst and a three-input Multiply, device against the host body
            const auto run_u = [&]<typename U>(bool dev, bool nary, U) {
                using F = typename U::value_type;
                std::vector<U> s(120001);
                for (std::size_t i = 0; i < s.size(); ++i) s[i] = U{F(1.5) + F(0.001) * F(i % 997) * (i % 2 ? F(1) : F(-1)), F(0.01) * F(1 + i % 13)};
                Graph g;
                auto& sink = g.emplaceBlock<testing::VectorSink<U>>();
                if (nary) {
                    auto& mul = g.emplaceBlock<blocks::math::Multiply<U>>(dev ? property_map{{"n_inputs", std::int64_t(3)}, {"compute_domain", "gpu:hip:0"s}} : property_map{{"n_inputs", std::int64_t(3)}});
                    for (std::size_t i = 0; i < 3; ++i) {
                        auto& src  = g.emplaceBlock<testing::VectorSource<U>>();
                        src.values.assign(s.begin() + static_cast<std::ptrdiff_t>(i * 7), s.begin() + static_cast<std::ptrdiff_t>(i * 7 + 100000));
                        if (!g.connect(src, "out"s, mul, "in#"s + std::to_string(i))) ++errors;
                    }
                    g.template connect<"out", "in">(mul, sink);
                    scheduler::Simple sched;
                    sched.exchange(std::move(g));
                    if (!sched.runAndWait()) ++errors;
                    if (dev && !mul._device_state) ++errors;
                    if (dev) hip::release(mul);
                    return sink._samples; // (copied while the scheduler still owns the graph: behind this block the sink is gone)
                } else {
                    auto& src  = g.emplaceBlock<testing::VectorSource<U>>();
                    src.values = s;
                    auto& div  = g.emplaceBlock<blocks::math::DivideConst<U>>(dev ? property_map{{"value", std::vector<double>{4.0, 0.5}}, {"compute_domain", "gpu:hip:0"s}} : property_map{{"value", std::vector<double>{4.0, 0.5}}});
                    g.template connect<"out", "in">(src, div);
                    g.template connect<"out", "in">(div, sink);
                    scheduler::Simple sched;
                    sched.exchange(std::move(g));
                    if (!sched.runAndWait()) ++errors;
                    if (dev && !div._device_state) ++errors;
                    if (dev) hip::release(div);
                    return sink._samples; // (a copy taken after the scheduler's destructor read freed memory: its first 32 bytes were the allocator's now and then)
                }
            };
            const auto cmp_u = [&](const char* what, const auto& d, const auto& h, std::size_t n, double tol) {
                double worst = d.size() == n && h.size() == n ? 0.0 : 1e30;
                for (std::size_t i = 0; i < n && worst < 1e29; ++i) {
                    if (d[i].value != h[i].value) {
                        worst = 1e30; // one IEEE operation per source operation: identical
                        std::size_t bad = 0, last = i;
                        for (std::size_t k = i; k < n; ++k)
                            if (d[k].value != h[k].value) { ++bad; last = k; }
                        std::printf("  %s: first differing element %zu (device {%g, %g}, host {%g, %g}), %zu differ, the last one %zu\n", what, i, double(d[i].value), double(d[i].uncertainty), double(h[i].value),
                                    double(h[i].uncertainty), bad, last);
                    }
                    worst = std::max(worst, std::abs(double(d[i].uncertainty) - double(h[i].uncertainty)) / std::abs(double(h[i].uncertainty)));
                }
                report(what, worst, tol);
            };
            cmp_u("DivideConst<UncertainValue<float>>", run_u(true, false, gr::UncertainValue<float>{}), run_u(false, false, gr::UncertainValue<float>{}), 120001, 4e-7);
            cmp_u("DivideConst<UncertainValue<double>>", run_u(true, false, gr::UncertainValue<double>{}), run_u(false, false, gr::UncertainValue<double>{}), 120001, 1e-15);
            cmp_u("Multiply<UncertainValue<float>> n_inputs = 3", run_u(true, true, gr::UncertainValue<float>{}), run_u(false, true, gr::UncertainValue<float>{}), 100000, 1e-6);
            cmp_u("Multiply<UncertainValue<double>> n_inputs = 3", run_u(true, true, gr::UncertainValue<double>{}), run_u(false, true, gr::UncertainValue<double>{}), 100000, 1e-15);
        }
        { // the FFT block: DataSets from the device equal the host body's (values to 1e-5 of the frame scale, identical descriptive part)
            std::vector<std::complex<float>> xc(1000 * 3 + 5);
            for (std::size_t i = 0; i < xc.size(); ++i) xc[i] = std::complex<float>(xf[2 * i], xf[2 * i + 1]) + std::polar(1.f, static_cast<float>(2 * std::numbers::pi * 0.1 * double(i % 1000)));
            const auto compare = [&](const char* what, const std::vector<DataSet<float>>& d, const std::vector<DataSet<float>>& h, std::size_t frames) {
                double worst = d.size() == frames && h.size() == frames ? 0.0 : 1e30;
                for (std::size_t f = 0; f < frames && worst < 1e29; ++f) {
                    if (d[f].axis_values != h[f].axis_values || d[f].signal_names != h[f].signal_names || d[f].signal_units != h[f].signal_units || d[f].extents != h[f].extents ||
                        d[f].meta_information != h[f].meta_information) worst = 1e30;
                    for (std::size_t i = 0; i < 4 && worst < 1e29; ++i) {
                        const auto dv = d[f].signalValues(i), hv = h[f].signalValues(i);
                        if (i == 1) continue; // phase wraps at +-pi and is noise in empty bins: pinned bin by bin against the oracle in tests/test_gpu_parity.py, by Re / Im here
                        worst = std::max(worst, max_rel(std::vector<float>(dv.begin(), dv.end()), std::vector<float>(hv.begin(), hv.end())));
                        const float span = std::max(std::abs(h[f].signal_ranges[i].min), std::abs(h[f].signal_ranges[i].max));
                        worst = std::max<double>(worst, std::abs(d[f].signal_ranges[i].min - h[f].signal_ranges[i].min) / span);
                        worst = std::max<double>(worst, std::abs(d[f].signal_ranges[i].max - h[f].signal_ranges[i].max) / span);
                    }
                    const auto sv = d[f].signalValues(1); // the device ranges are the min / max of the device's own phase values
                    if (d[f].signal_ranges[1].min != *std::min_element(sv.begin(), sv.end()) || d[f].signal_ranges[1].max != *std::max_element(sv.begin(), sv.end())) worst = 1e30;
                }
                report(what, worst, 2e-5);
            };
            for (const std::int64_t N : {std::int64_t(256), std::int64_t(1000)}) { // power of two and Bluestein
                using B = blocks::fft::FFT<std::complex<float>>;
                const property_map cfg{{"fftSize", N}, {"window", N == 256 ? "Hann"s : "BlackmanHarris"s}, {"sample_rate", 1000.0}, {"signal_name", "ch0"s}};
                compare(N == 256 ? "FFT<complex<float>> 256 Hann" : "FFT<complex<float>> 1000 B-Harris", run_one<B, std::complex<float>, DataSet<float>>(cfg, xc, true, errors),
                        run_one<B, std::complex<float>, DataSet<float>>(cfg, xc, false, errors), xc.size() / static_cast<std::size_t>(N));
            }
            using BR = blocks::fft::FFT<float>;
            const property_map cfg{{"fftSize", std::int64_t(512)}, {"window", "Hamming"s}, {"outputInDb", true}};
            compare("FFT<float> 512 Hamming dB", run_one<BR, float, DataSet<float>>(cfg, xf, true, errors), run_one<BR, float, DataSet<float>>(cfg, xf, false, errors), xf.size() / 512);
        }
        { // planner over resampling stages: MultiplyConst -> BasicDecimatingFilter(FIR, /4) -> Decimator(/3) -> iir_filter, one run, one stream (the gain in the filter's taps)
            std::vector<float> got[2];
            for (int dev = 1; dev >= 0; --dev) {
                Graph g;
                const auto dom = [&](property_map m) { if (dev) m["compute_domain"] = "gpu:hip:0"s; return m; };
                auto& src  = g.emplaceBlock<testing::VectorSource<float>>();
                src.values = xf;
                auto& mul  = g.emplaceBlock<blocks::math::MultiplyConst<float>>(dom({{"value", 0.5}}));
                auto& bdf  = g.emplaceBlock<filter::BasicDecimatingFilter<float>>(dom({{"filter_type", "FIR"s}, {"filter_order", std::int64_t(4)}, {"f_low", 50.0}, {"sample_rate", 1000.0}, {"fir_design_method", "Hamming"s}, {"decimate", std::int64_t(4)}}));
                auto& dec  = g.emplaceBlock<filter::Decimator<float>>(dom({{"decim", std::int64_t(3)}}));
                auto& iir  = g.emplaceBlock<filter::iir_filter<float>>(dom({{"b", bq_b}, {"a", bq_a}}));
                auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
                g.connect<"out", "in">(src, mul);
                g.connect<"out", "in">(mul, bdf);
                g.connect<"out", "in">(bdf, dec);
                g.connect<"out", "in">(dec, iir);
                g.connect<"out", "in">(iir, sink);
                hip::DeviceRun* the_run = nullptr;
                if (dev) {
                    const auto runs = hip::plan(g, 2, 0); // (default-size edges on purpose: several launches, so that the pipelining shows)
                    the_run = runs.empty() ? nullptr : runs[0];
                    std::printf("planner (resampling): %zu run:%s%s\n", runs.size(), runs.empty() ? "" : " ", runs.empty() ? "" : std::string(runs[0]->description()).c_str());
                    if (runs.size() != 1 || runs[0]->description() != "basic_fir_decim[pre: mul] -> decimator -> iir_f32" || runs[0]->out_count(12) != 1) ++errors;
                }
                scheduler::Simple sched;
                sched.exchange(std::move(g));
                if (const auto r = sched.runAndWait(); !r) { std::cerr << "resampling run: " << r.error().message << "\n"; ++errors; }
                got[dev] = sink._samples;
                if (dev) { // work() only queues: chunk c + 1 is copied in while chunk c computes and chunk c - 1 is copied out
                    const std::size_t ov = the_run ? the_run->overlapped_chunks() : 0;
                    std::printf("pipelined run: %zu of %zu launches queued while an earlier chunk was still in flight\n", ov, the_run ? the_run->launches() / 3 : 0);
                    // (reported, not asserted: whether a chunk is still in flight when the next one is queued is a matter of timing on five small chunks; the
                    //  pipelining itself is measured by bench_host_feed: profiles/r02_host_feed.txt)
                }
            }
            report("planned run with two rate changes", got[1].size() == xf.size() / 12 ? max_rel(got[1], got[0]) : 1e30, 1e-5);
        }
    }
    { // 5b. merge API on the device: Merge<MultiplyConst, FeedbackMerge<Adder, MultiplyConst>> (the reference benchmark's IIR low-pass) is ONE first-order
      //     section for the scan kernel; a Merge of a gain and a filter is the filter with the gain in its taps (5c: every other fusable pair)
        using blocks::math::MultiplyConst;
        using IIRChain = gr::Merge<MultiplyConst<float>, "out", gr::FeedbackMerge<Adder<>, "out", MultiplyConst<float>, "out", "in2">, "in1">;
        std::vector<float> xs(400000);
        std::uint32_t      lcg = 99u;
        for (auto& v : xs) { lcg = lcg * 1664525u + 1013904223u; v = static_cast<float>(static_cast<std::int32_t>(lcg >> 8) % 2001 - 1000) / 1000.f; }
        const property_map cfg{{"leftBlock.value", double(0.3f)}, {"rightBlock.feedback.value", double(1.0f - 0.3f)}};
        const auto d = run_one<IIRChain, float, float>(cfg, xs, true, errors), h = run_one<IIRChain, float, float>(cfg, xs, false, errors);
        const double e1 = max_rel(d, h);
        IIRChain probe;
        probe.applySettings(cfg);
        const auto st = hip::Kernel<IIRChain>::make_stage(probe);
        std::printf("merge IIR low-pass (FeedbackMerge) on the device: stage '%s', max rel err %.3g%s\n", std::string(st->kind()).c_str(), e1, e1 <= 1e-5 ? "" : "  FAILED");
        if (!(e1 <= 1e-5) || st->kind() != "iir_f32") ++errors;
        // the same low-pass with the feedback gain decomposed into a SplitMergeCombine of two gains (bm_MergeApi.cpp:62-64): still one first-order section
        using IIRChainSplitMerge = gr::Merge<MultiplyConst<float>, "out", gr::FeedbackMerge<Adder<>, "out", gr::SplitMergeCombine<MultiplyConst<float>, MultiplyConst<float>>, "out", "in2">, "in1">;
        const property_map cfg_sm{{"leftBlock.value", double(0.3f)}, {"rightBlock.feedback.path0.value", 1.0}, {"rightBlock.feedback.path1.value", double(-0.3f)}};
        const double e_sm = max_rel(run_one<IIRChainSplitMerge, float, float>(cfg_sm, xs, true, errors), h);
        IIRChainSplitMerge probe_sm;
        probe_sm.applySettings(cfg_sm);
        std::printf("merge IIR low-pass (SplitMergeCombine feedback) on the device: stage '%s', max rel err %.3g%s\n", std::string(hip::Kernel<IIRChainSplitMerge>::make_stage(probe_sm)->kind()).c_str(), e_sm,
                    e_sm <= 1e-5 ? "" : "  FAILED");
        if (!(e_sm <= 1e-5) || hip::Kernel<IIRChainSplitMerge>::make_stage(probe_sm)->kind() != "iir_f32") ++errors;
        // a general SplitMergeCombine: x -> fir(x) - 0.25 x, every path a device stage, the signed sum on the math kernels
        using Split = gr::SplitMergeCombine<gr::OutputSigns<+1.0f, -1.0f>, filter::fir_filter<float>, MultiplyConst<float>>;
        const property_map cfg_sp{{"path0.b", std::vector<double>{0.5, 0.25, 0.25}}, {"path1.value", 0.25}};
        const double e_sp = max_rel(run_one<Split, float, float>(cfg_sp, xs, true, errors), run_one<Split, float, float>(cfg_sp, xs, false, errors));
        Split probe_sp;
        probe_sp.applySettings(cfg_sp);
        std::printf("SplitMergeCombine on the device: stage '%s', max rel err %.3g%s\n", std::string(hip::Kernel<Split>::make_stage(probe_sp)->kind()).c_str(), e_sp, e_sp <= 1e-5 ? "" : "  FAILED");
        if (!(e_sp <= 1e-5)) ++errors;
        using Two = gr::Merge<MultiplyConst<float>, "out", filter::fir_filter<float>, "in">;
        const property_map cfg2{{"leftBlock.value", 2.0}, {"rightBlock.b", std::vector<double>{0.5, 0.25, 0.25}}};
        const double e2 = max_rel(run_one<Two, float, float>(cfg2, xs, true, errors), run_one<Two, float, float>(cfg2, xs, false, errors));
        Two probe2;
        probe2.applySettings(cfg2);
        std::printf("merge MultiplyConst -> fir_filter on the device: stage '%s', max rel err %.3g%s\n", std::string(hip::Kernel<Two>::make_stage(probe2)->kind()).c_str(), e2, e2 <= 1e-5 ? "" : "  FAILED");
        if (!(e2 <= 1e-5)) ++errors;
    }
    { // 5c. kernel-level fusion, the run-time Merge<> (BlockMerging.hpp:126-240): adjacent per-sample blocks are ONE launch, and they ride in the launch of the
      //     filter next to them -- no intermediate stream in HBM
        using namespace blocks::math;
        std::vector<float> xs(300000);
        std::uint32_t      lcg = 7u;
        for (auto& v : xs) { lcg = lcg * 1664525u + 1013904223u; v = static_cast<float>(static_cast<std::int32_t>(lcg >> 8) % 2001 - 1000) / 250.f; }
        // (i) the reference's merged benchmark chains (core/benchmarks/bm_MergeApi.cpp:174: mult -> div -> add, and that ten times over): one program each,
        //     bit-identical to the merged block on the host (the same IEEE operations in the same order)
        using Chain1  = gr::Merge<MultiplyConst<float>, "out", gr::Merge<DivideConst<float>, "out", AddConst<float>, "in">, "in">;
        using Chain2  = gr::Merge<Chain1, "out", Chain1, "in">;
        using Chain4  = gr::Merge<Chain2, "out", Chain2, "in">;
        using Chain8  = gr::Merge<Chain4, "out", Chain4, "in">;
        using Chain10 = gr::Merge<Chain8, "out", Chain2, "in">;
        for (const double value : {2.0, 3.0}) { // 2: the quotient is a product with 0.5, bit for bit; 3: a true division
            const property_map cfg{{"value", value}}; // a flat key reaches every part that has such a setting
            const auto d1 = run_one<Chain1, float, float>(cfg, xs, true, errors), h1 = run_one<Chain1, float, float>(cfg, xs, false, errors);
            const auto d10 = run_one<Chain10, float, float>(cfg, xs, true, errors), h10 = run_one<Chain10, float, float>(cfg, xs, false, errors);
            Chain1 p1;
            Chain10 p10;
            p1.applySettings(cfg);
            p10.applySettings(cfg);
            const auto s1 = hip::Kernel<Chain1>::make_stage(p1);
            const auto s10 = hip::Kernel<Chain10>::make_stage(p10);
            const bool same1 = d1.size() == h1.size() && std::memcmp(d1.data(), h1.data(), d1.size() * sizeof(float)) == 0;
            const bool same10 = d10.size() == h10.size() && std::memcmp(d10.data(), h10.data(), d10.size() * sizeof(float)) == 0;
            std::printf("merge mult->div->add (value %g) on the device: stage '%s', %s the host block; (mult->div->add)^10: stage '%s', %s\n", value, std::string(s1->kind()).c_str(),
                        same1 ? "bit-identical to" : "DIFFERS from", std::string(s10->kind()).c_str(), same10 ? "bit-identical" : "DIFFERS");
            if (!same1 || !same10 || s1->kind() != "ewise[mul,div,add]" || s10->kind() != "ewise[30 ops]") ++errors;
        }
        { // the integer chain wraps like the C++ blocks: bit-exact
            using IChain = gr::Merge<MultiplyConst<std::int32_t>, "out", gr::Merge<DivideConst<std::int32_t>, "out", AddConst<std::int32_t>, "in">, "in">;
            std::vector<std::int32_t> xi(100000);
            for (auto& v : xi) { lcg = lcg * 1664525u + 1013904223u; v = static_cast<std::int32_t>(lcg); }
            const property_map cfg{{"leftBlock.value", std::int64_t(77777)}, {"rightBlock.leftBlock.value", std::int64_t(-13)}, {"rightBlock.rightBlock.value", std::int64_t(2000000000)}};
            const bool same = run_one<IChain, std::int32_t, std::int32_t>(cfg, xi, true, errors) == run_one<IChain, std::int32_t, std::int32_t>(cfg, xi, false, errors);
            std::printf("merge mult->div->add <int32> on the device: %s the host block\n", same ? "bit-identical to" : "DIFFERS from");
            if (!same) ++errors;
        }
        { // (ii) the same chain as three graph blocks: the planner makes ONE stage of them (one launch per chunk, 8 B of HBM traffic per sample instead of 24)
            std::vector<float> got[2];
            std::size_t        stages = 0, launches = 0;
            std::string        desc;
            for (int dev = 1; dev >= 0; --dev) {
                Graph g;
                const auto dom = [&](property_map m) { if (dev) m["compute_domain"] = "gpu:hip:0"s; return m; };
                auto& src  = g.emplaceBlock<testing::VectorSource<float>>();
                src.values = xs;
                auto& mul  = g.emplaceBlock<MultiplyConst<float>>(dom({{"value", 3.0}}));
                auto& div  = g.emplaceBlock<DivideConst<float>>(dom({{"value", 7.0}}));
                auto& add  = g.emplaceBlock<AddConst<float>>(dom({{"value", -1.0}}));
                auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
                if (!g.connect<"out", "in">(src, mul) || !g.connect<"out", "in">(mul, div) || !g.connect<"out", "in">(div, add) || !g.connect<"out", "in">(add, sink)) ++errors;
                hip::DeviceRun* run = nullptr;
                if (dev) {
                    const auto runs = hip::plan(g);
                    if (runs.size() != 1) { ++errors; break; }
                    run    = runs[0];
                    stages = run->stages().size();
                    desc   = std::string(run->description());
                }
                scheduler::Simple sched;
                sched.exchange(std::move(g));
                if (const auto r = sched.runAndWait(); !r) { std::cerr << "fused math run: " << r.error().message << "\n"; ++errors; }
                got[dev] = sink._samples;
                if (run) launches = run->launches();
            }
            const bool same = got[1].size() == xs.size() && got[1] == got[0];
            std::printf("planner (math chain): 1 run: %s  (%zu stage, %zu launch%s), output %s the host graph\n", desc.c_str(), stages, launches, launches == 1 ? "" : "es", same ? "bit-identical to" : "DIFFERS from");
            if (stages != 1 || desc != "ewise[mul,div,add]" || !same) ++errors;
        }
        { // (iii) the channeliser: Rotator -> BasicDecimatingFilter<complex<float>> (designed real taps on complex data) -> PowerSpectrum.  The rotator is the
          //       filter's load hook: two launches per chunk (filter, transform), and the only intermediate in HBM is the DECIMATED stream.  Checked against the oracle
          //       by tests/test_host_cpp.py (the host Rotator accumulates its phase in float and drifts away from the closed form over a long stream)
            Graph g;
            auto& src  = g.emplaceBlock<testing::VectorSource<std::complex<float>>>({{"n_samples_max", std::int64_t(8 * 256 * 24)}});
            src.values = x;
            auto& rot  = g.emplaceBlock<Rotator<std::complex<float>>>({{"phase_increment", 0.3}, {"initial_phase", 0.25}, {"compute_domain", "gpu:hip:0"s}});
            auto& bdf  = g.emplaceBlock<filter::BasicDecimatingFilter<std::complex<float>>>({{"filter_type", "FIR"s}, {"filter_order", std::int64_t(4)}, {"f_low", 40.0}, {"sample_rate", 1000.0},
                                                                                             {"fir_design_method", "Hamming"s}, {"decimate", std::int64_t(8)}, {"compute_domain", "gpu:hip:0"s}});
            auto& spec = g.emplaceBlock<blocks::fft::PowerSpectrum<std::complex<float>>>({{"fftSize", std::int64_t(256)}, {"window", "Hann"s}, {"compute_domain", "gpu:hip:0"s}});
            auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
            if (!g.connect<"out", "in">(src, rot) || !g.connect<"out", "in">(rot, bdf) || !g.connect<"out", "in">(bdf, spec) || !g.connect<"out", "in">(spec, sink)) ++errors;
            const auto runs = hip::plan(g);
            scheduler::Simple sched;
            sched.exchange(std::move(g));
            if (const auto r = sched.runAndWait(); !r) { std::cerr << "channeliser: " << r.error().message << "\n"; ++errors; }
            if (runs.size() == 1) {
                std::printf("planner (channeliser): 1 run: %s  (%zu stages, %zu launches for %zu samples in)\n", std::string(runs[0]->description()).c_str(), runs[0]->stages().size(), runs[0]->launches(),
                            std::size_t(8 * 256 * 24));
                if (runs[0]->stages().size() != 2 || runs[0]->description() != "basic_fir_decim[pre: rot] -> power_spectrum_c32") ++errors;
            } else ++errors;
            if (sink._samples.size() != 256 * 24) ++errors;
            dump(out + "_channeliser.bin", sink._samples);
            dump(out + "_channeliser_taps.bin", bdf._design.taps);
        }
        { // (iv) fir_filter -> Decimator is the polyphase decimating FIR (only the kept outputs are computed), with the gain in front of it in its taps
            std::vector<float> got[2];
            std::string        desc;
            for (int dev = 1; dev >= 0; --dev) {
                Graph g;
                const auto dom = [&](property_map m) { if (dev) m["compute_domain"] = "gpu:hip:0"s; return m; };
                auto& src  = g.emplaceBlock<testing::VectorSource<float>>();
                src.values = xs;
                auto& mul  = g.emplaceBlock<MultiplyConst<float>>(dom({{"value", 0.5}}));
                auto& fir  = g.emplaceBlock<filter::fir_filter<float>>(dom({{"b", std::vector<double>{0.1, 0.2, 0.4, 0.2, 0.1, 0.05, -0.05}}}));
                auto& dec  = g.emplaceBlock<filter::Decimator<float>>(dom({{"decim", std::int64_t(5)}}));
                auto& add  = g.emplaceBlock<AddConst<float>>(dom({{"value", 2.0}}));
                auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
                if (!g.connect<"out", "in">(src, mul) || !g.connect<"out", "in">(mul, fir) || !g.connect<"out", "in">(fir, dec) || !g.connect<"out", "in">(dec, add) || !g.connect<"out", "in">(add, sink)) ++errors;
                if (dev) {
                    const auto runs = hip::plan(g);
                    if (runs.size() != 1) { ++errors; break; }
                    desc = std::string(runs[0]->description());
                }
                scheduler::Simple sched;
                sched.exchange(std::move(g));
                if (const auto r = sched.runAndWait(); !r) { std::cerr << "fir -> decimator: " << r.error().message << "\n"; ++errors; }
                got[dev] = sink._samples;
            }
            const double e = got[1].size() == xs.size() / 5 ? max_rel(got[1], got[0]) : 1e30;
            std::printf("planner (gain -> fir -> Decimator -> add): 1 run: %s, max rel err vs the host graph %.3g%s\n", desc.c_str(), e, e <= 1e-5 ? "" : "  FAILED");
            if (!(e <= 1e-5) || desc != "fir_f32/5[pre: mul; post: add]") ++errors;
        }
        { // (v) per-sample blocks around a Decimator ride behind it in its launch (they commute with dropping samples); float blocks behind a PowerSpectrum ride in the
          //     transform's launch.  Against the same graphs on the host
            std::vector<std::int32_t> xi(90001);
            for (auto& v : xi) { lcg = lcg * 1664525u + 1013904223u; v = static_cast<std::int32_t>(lcg >> 4); }
            std::vector<std::int32_t> gi[2];
            std::vector<float>        gs[2];
            std::string               d1, d2;
            for (int dev = 1; dev >= 0; --dev) {
                Graph g;
                const auto dom = [&](property_map m) { if (dev) m["compute_domain"] = "gpu:hip:0"s; return m; };
                auto& src  = g.emplaceBlock<testing::VectorSource<std::int32_t>>();
                src.values = xi;
                auto& add  = g.emplaceBlock<AddConst<std::int32_t>>(dom({{"value", std::int64_t(12345)}}));
                auto& dec  = g.emplaceBlock<filter::Decimator<std::int32_t>>(dom({{"decim", std::int64_t(7)}}));
                auto& mul  = g.emplaceBlock<MultiplyConst<std::int32_t>>(dom({{"value", std::int64_t(-3)}}));
                auto& sink = g.emplaceBlock<testing::VectorSink<std::int32_t>>();
                auto& csrc = g.emplaceBlock<testing::VectorSource<std::complex<float>>>({{"n_samples_max", std::int64_t(40 * 256)}});
                csrc.values = x;
                auto& spec = g.emplaceBlock<blocks::fft::PowerSpectrum<std::complex<float>>>(dom({{"fftSize", std::int64_t(256)}, {"window", "Hann"s}}));
                auto& nrm  = g.emplaceBlock<DivideConst<float>>(dom({{"value", 65536.0}}));
                auto& off  = g.emplaceBlock<AddConst<float>>(dom({{"value", 0.5}}));
                auto& ssnk = g.emplaceBlock<testing::VectorSink<float>>();
                if (!g.connect<"out", "in">(src, add) || !g.connect<"out", "in">(add, dec) || !g.connect<"out", "in">(dec, mul) || !g.connect<"out", "in">(mul, sink) ||
                    !g.connect<"out", "in">(csrc, spec) || !g.connect<"out", "in">(spec, nrm) || !g.connect<"out", "in">(nrm, off) || !g.connect<"out", "in">(off, ssnk)) ++errors;
                if (dev) {
                    const auto runs = hip::plan(g);
                    if (runs.size() != 2) { ++errors; break; }
                    d1 = std::string(runs[0]->description());
                    d2 = std::string(runs[1]->description());
                }
                scheduler::Simple sched;
                sched.exchange(std::move(g));
                if (const auto r = sched.runAndWait(); !r) { std::cerr << "decimator / spectrum hooks: " << r.error().message << "\n"; ++errors; }
                gi[dev] = sink._samples;
                gs[dev] = ssnk._samples;
            }
            const double e = gs[1].size() == 40 * 256 && gs[0].size() == gs[1].size() ? max_rel(gs[1], gs[0]) : 1e30;
            std::printf("planner (blocks around a Decimator, behind a PowerSpectrum): [%s] %s the host graph; [%s] max rel err %.3g%s\n", d1.c_str(), gi[1] == gi[0] && !gi[1].empty() ? "bit-identical to" : "DIFFERS from",
                        d2.c_str(), e, e <= 1e-5 ? "" : "  FAILED");
            if (gi[1] != gi[0] || gi[1].size() != xi.size() / 7 /* whole decimation groups: input_chunk_size = decim */ || !(e <= 1e-5) || d1 != "decimator[post: add,mul]" || d2 != "power_spectrum_c32[post: div,add]") ++errors;
        }
    }
    { // 5b. the multi-channel graph of BASELINE configs[4] in the C++ API: every channel its own planned device run (fir_filter -> PowerSpectrum fused), channel c on
      //     device c mod <devices> (one here; the runs set their device on every work() call), the fan-in combiner math::Add<float> with n_inputs = channels
      //     (Math.hpp:73-108) on device 0.  Against the same graph on the host.
        int n_dev = 1;
        (void)gr4hip_device_count(&n_dev);
        constexpr std::size_t kCh = 4;
        std::vector<float> sums[2];
        std::size_t        n_runs = 0;
        for (int dev = 1; dev >= 0; --dev) {
            Graph g;
            auto& add = g.emplaceBlock<blocks::math::Add<float>>(dev ? property_map{{"n_inputs", std::int64_t(kCh)}, {"compute_domain", "gpu:hip:0"s}} : property_map{{"n_inputs", std::int64_t(kCh)}});
            for (std::size_t c = 0; c < kCh; ++c) {
                const std::string dom = "gpu:hip:" + std::to_string(c % static_cast<std::size_t>(std::max(1, n_dev)));
                auto& src  = g.emplaceBlock<testing::VectorSource<std::complex<float>>>({{"n_samples_max", std::int64_t(6 * N)}});
                src.values.assign(x.begin() + static_cast<std::ptrdiff_t>(37 * c), x.end()); // the channels see different streams
                auto& fir  = g.emplaceBlock<filter::fir_filter<std::complex<float>>>(dev ? property_map{{"b", tapsd}, {"compute_domain", dom}} : property_map{{"b", tapsd}});
                auto& spec = g.emplaceBlock<blocks::fft::PowerSpectrum<std::complex<float>>>(
                    dev ? property_map{{"fftSize", std::int64_t(N)}, {"window", "Hann"s}, {"compute_domain", dom}} : property_map{{"fftSize", std::int64_t(N)}, {"window", "Hann"s}});
                if (!g.connect<"out", "in">(src, fir) || !g.connect<"out", "in">(fir, spec) || !g.connect(spec, "out"s, add, "in#"s + std::to_string(c))) ++errors;
            }
            auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
            if (!g.connect<"out", "in">(add, sink)) ++errors;
            if (dev) n_runs = hip::plan(g).size();
            scheduler::Simple sched;
            sched.exchange(std::move(g));
            if (const auto r = sched.runAndWait(); !r) { std::cerr << "fan-in graph: " << r.error().message << "\n"; ++errors; }
            sums[dev] = sink._samples;
        }
        const double e = sums[1].size() == 6 * N && sums[0].size() == sums[1].size() ? max_rel(sums[1], sums[0]) : 1e30;
        std::printf("fan-in graph: %zu channels, %zu fused runs on %d device(s), Add<float> n_inputs = %zu on the device: max rel err vs the host graph %.3g%s\n", kCh, n_runs, n_dev, kCh, e,
                    e <= 1e-5 ? "" : "  FAILED");
        if (n_runs != kCh || !(e <= 1e-5)) ++errors;
    }
    { // 6. tags through a fused device run: launches split at tags, "gr:" keys forwarded at the first output sample with gr:sample_rate scaled by the run's rate
      //    change, settings-by-tag reaches the member block and rebuilds only its stage (the FIR keeps its history)
        std::vector<float> xs(120000);
        for (std::size_t i = 0; i < xs.size(); ++i) xs[i] = static_cast<float>(std::sin(0.01 * double(i)));
        std::vector<float> got[2];
        std::vector<Tag>   tags[2];
        for (int dev = 1; dev >= 0; --dev) {
            Graph g;
            const auto dom = [&](property_map m) { if (dev) m["compute_domain"] = "gpu:hip:0"s; return m; };
            auto& src  = g.emplaceBlock<testing::VectorSource<float>>();
            src.values = xs;
            src._tags  = {{0, {{"gr:sample_rate", 1000.f}}}, {40000, {{"gr:sample_rate", 1000.f}, {"gr:value", 3.0}, {"note", "dropped"s}}}};
            auto& mul  = g.emplaceBlock<blocks::math::MultiplyConst<float>>(dom({{"value", 0.5}}));
            auto& bdf  = g.emplaceBlock<filter::BasicDecimatingFilter<float>>(dom({{"filter_type", "FIR"s}, {"filter_order", std::int64_t(4)}, {"f_low", 50.0}, {"sample_rate", 1000.0}, {"fir_design_method", "Hamming"s}, {"decimate", std::int64_t(4)}}));
            auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
            g.connect<"out", "in">(src, mul);
            g.connect<"out", "in">(mul, bdf);
            g.connect<"out", "in">(bdf, sink);
            hip::DeviceRun* run = nullptr;
            if (dev) {
                const auto runs = hip::plan(g);
                if (runs.size() != 1) { ++errors; break; }
                run = runs[0];
            }
            scheduler::Simple sched;
            sched.exchange(std::move(g));
            if (const auto r = sched.runAndWait(); !r) { std::cerr << "tagged run: " << r.error().message << "\n"; ++errors; }
            got[dev]  = sink._samples;
            tags[dev] = sink._tags;
            if (dev) {
                std::printf("tags through the device run: %zu forwarded, %zu stage rebuilt, %zu launches\n", run->tags_forwarded(), run->stages_rebuilt(), run->launches());
                if (run->tags_forwarded() != 2 || run->stages_rebuilt() != 1) ++errors;
            }
        }
        const std::vector<Tag> want{{0, {{"gr:sample_rate", 250.f}}}, {10000, {{"gr:sample_rate", 250.f}, {"gr:value", 3.0}}}};
        double worst = got[1].size() == xs.size() / 4 && got[0].size() == got[1].size() ? 0.0 : 1e30;
        for (std::size_t i = 0; i < got[1].size() && worst < 1e29; ++i) worst = std::max(worst, double(std::abs(got[1][i] - got[0][i])));
        std::printf("tags: device run %s, host graph %s, streams differ by %.3g\n", tags[1] == want ? "forwarded {0: 250 Hz, 10000: 250 Hz + gr:value}" : "WRONG TAGS",
                    tags[0] == want ? "the same" : "DIFFERENT", worst);
        if (tags[1] != want || tags[0] != want || !(worst <= 2e-5)) ++errors;
        // after the tag the run multiplies by 3 instead of 0.5 with the SAME filter state: y_after = FIR(3 x) continues FIR(0.5 x) without a transient
        // other than the one of the gain step itself; compare against a host rendering of exactly that
        {
            filter::BasicDecimatingFilter<float> ref;
            ref.applySettings({{"filter_type", "FIR"s}, {"filter_order", std::int64_t(4)}, {"f_low", 50.0}, {"sample_rate", 1000.0}, {"fir_design_method", "Hamming"s}, {"decimate", std::int64_t(4)}});
            std::vector<float> scaled(xs.size()), y(xs.size() / 4);
            for (std::size_t i = 0; i < xs.size(); ++i) scaled[i] = xs[i] * (i < 40000 ? 0.5f : 3.f);
            (void)ref.processBulk(scaled, y);
            double w2 = 0;
            for (std::size_t i = 0; i < y.size() && i < got[1].size(); ++i) w2 = std::max(w2, double(std::abs(got[1][i] - y[i])));
            std::printf("tags: gain step at the tagged sample, filter state kept: max diff %.3g\n", w2);
            if (!(w2 <= 2e-5)) ++errors;
        }
    }
    { // 6b. tags INSIDE a chunk the run cannot split (a decimation group of 100 samples): every tag of the chunk is merged and forwarded at the chunk's
      //     first output sample, and a setting among them is applied -- on the device run exactly as in the host graph
        std::vector<Tag>   tags[2];
        std::vector<float> got[2];
        for (int dev = 1; dev >= 0; --dev) {
            Graph g;
            const auto dom = [&](property_map m) { if (dev) m["compute_domain"] = "gpu:hip:0"s; return m; };
            auto& src  = g.emplaceBlock<testing::VectorSource<float>>({{"n_samples_max", std::int64_t(500)}});
            src.values = {1.f, 2.f, 3.f};
            src._tags  = {{0, {{"gr:sample_rate", 1000.f}}}, {250, {{"gr:sample_rate", 2000.f}, {"gr:trigger_name", "a"s}}}, {260, {{"gr:trigger_name", "b"s}, {"gr:value", 4.0}}}, {400, {{"gr:trigger_name", "c"s}}}};
            auto& mul  = g.emplaceBlock<blocks::math::MultiplyConst<float>>(dom({{"value", 0.5}}));
            auto& dec  = g.emplaceBlock<filter::Decimator<float>>(dom({{"decim", std::int64_t(100)}}));
            auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
            g.connect<"out", "in">(src, mul);
            g.connect<"out", "in">(mul, dec);
            g.connect<"out", "in">(dec, sink);
            if (dev && hip::plan(g).size() != 1) { ++errors; break; }
            scheduler::Simple sched;
            sched.exchange(std::move(g));
            if (const auto r = sched.runAndWait(); !r) { std::cerr << "inner-tag run: " << r.error().message << "\n"; ++errors; }
            tags[dev] = sink._tags;
            got[dev]  = sink._samples;
        }
        // the host graph splits MultiplyConst's chunks at every tag, so it sees 250 and 260 separately and the Decimator merges them; the run sees them merged
        const std::vector<Tag> want{{0, {{"gr:sample_rate", 10.f}}}, {2, {{"gr:sample_rate", 20.f}, {"gr:trigger_name", "b"s}, {"gr:value", 4.0}}}, {4, {{"gr:trigger_name", "c"s}}}};
        std::printf("tags inside a decimation group: device run %s, host graph %s\n", tags[1] == want ? "forwarded all of them merged" : "WRONG TAGS", tags[0] == want ? "the same" : "DIFFERENT");
        if (tags[1] != want || tags[0] != want || got[1].size() != 5 || got[0].size() != 5) ++errors;
        // samples 0 and 100 carry the old gain in both; from the chunk that holds the tag on, the run applies the new value to the whole chunk
        if (got[1].size() == 5 && !(got[1][0] == 0.5f && got[1][1] == 1.f && got[1][3] == 4.f && got[1][4] == 8.f)) ++errors; // x[0]=1, x[100]=2, x[300]=1, x[400]=2
    }
    // ------------------------------------------------------------------ settings-by-tag on a LONE device block (no DeviceRun: the per-block seam)
    // a stage is built once and must follow the block's settings: a gain step by tag on MultiplyConst, new taps by tag on fir_filter (history kept)
    {
        const std::size_t   n = 60000, at = 23456;
        std::vector<float> xs(n);
        for (std::size_t i = 0; i < n; ++i) xs[i] = static_cast<float>(std::sin(0.013 * double(i)) + 0.25 * std::cos(0.4 * double(i)));
        std::vector<float> b1(33, 1.f / 33.f), b2(33);
        for (std::size_t k = 0; k < b2.size(); ++k) b2[k] = (k % 2 ? -1.f : 1.f) / 33.f;
        for (int which = 0; which < 2; ++which) {
            std::vector<float> got[2];
            for (int dev = 0; dev < 2; ++dev) {
                Graph g;
                auto& src  = g.emplaceBlock<testing::VectorSource<float>>();
                src.values = xs;
                auto& sink = g.emplaceBlock<testing::VectorSink<float>>();
                bool  ok   = true;
                void* took = nullptr;
                scheduler::Simple sched; // owns the graph from exchange() on: must outlive the reads of the sink below
                if (which == 0) {
                    src._tags = {{at, {{"value", 3.0}}}};
                    property_map m{{"value", 0.5}};
                    if (dev) m["compute_domain"] = "gpu:hip:0"s;
                    auto& blk = g.emplaceBlock<blocks::math::MultiplyConst<float>>(m);
                    ok        = bool(g.connect<"out", "in">(src, blk)) && bool(g.connect<"out", "in">(blk, sink));
                    sched.exchange(std::move(g));
                    if (const auto r = sched.runAndWait(); !r) ok = false;
                    took = blk._device_state.get();
                } else {
                    src._tags = {{at, {{"b", b2}}}};
                    property_map m{{"b", b1}};
                    if (dev) m["compute_domain"] = "gpu:hip:0"s;
                    auto& blk = g.emplaceBlock<filter::fir_filter<float>>(m);
                    ok        = bool(g.connect<"out", "in">(src, blk)) && bool(g.connect<"out", "in">(blk, sink));
                    sched.exchange(std::move(g));
                    if (const auto r = sched.runAndWait(); !r) ok = false;
                    took = blk._device_state.get();
                }
                if (!ok || (dev && !took)) ++errors;
                got[dev] = sink._samples;
            }
            double worst = got[0].size() == n && got[1].size() == n ? 0.0 : 1e30;
            for (std::size_t i = 0; i < n && worst < 1e29; ++i) worst = std::max(worst, double(std::abs(got[1][i] - got[0][i])));
            const bool stepped = got[0].size() == n && (which == 1 || (got[0][at - 1] == xs[at - 1] * 0.5f && got[0][at] == xs[at] * 3.f));
            std::printf("settings-by-tag on a lone device block (%s): device vs host max diff %.3g%s\n", which == 0 ? "MultiplyConst value" : "fir_filter taps, history kept", worst,
                        stepped ? "" : "  (host step missing)");
            if (!(worst <= 2e-6) || !stepped) ++errors;
        }
    }
    { // a stage that fails in the middle of the stream: the run reports ERROR, the chunk whose launch failed and the chunk whose copy was already on its way go back
      // to the input edge untouched, nothing is published twice or lost, and the stream completes once the stage works again (page-locked and ordinary edges)
        struct FlakyStage final : hip::Stage {
            hip::MathConstStage<float, GR4HIP_MUL> inner{3.f};
            int calls = 0, fail_at;
            explicit FlakyStage(int fail_at_) : fail_at(fail_at_) { in_bytes = out_bytes = 4; }
            std::string_view kind() const override { return "flaky"; }
            int enqueue(const void* in, std::size_t n, void* o, std::size_t* n_out, gr4hip_stream_t st) override {
                if (++calls == fail_at) return GR4HIP_RUNTIME_ERROR;
                return inner.enqueue(in, n, o, n_out, st);
            }
        };
        for (int variant = 0; variant < 4; ++variant) { // edges that hold whole chunks (2^23 items), then the reference's default 65 536-item edges: gathered in the device ring (round 6) --
            const int pinned = variant & 1;              // there the chunk whose launch failed stays in the ring and is launched again by the next call (its spans went back long ago)
            hip::register_provider();
            std::pmr::memory_resource* mr = pinned ? hip::pinned_resource() : std::pmr::get_default_resource();
            const std::size_t cap = variant < 2 ? std::size_t(1) << 23 : std::size_t(1) << 16, total = std::size_t(5) << 22;
            auto in  = std::make_shared<EdgeBuffer<float>>(cap, mr);
            auto outb = std::make_shared<EdgeBuffer<float>>(cap, mr);
            std::vector<std::unique_ptr<hip::Stage>> st;
            st.push_back(std::make_unique<FlakyStage>(3));
            hip::DeviceRun run(std::move(st), in, outb, ComputeDomain::parse("gpu:hip:0"));
            std::size_t fed = 0, got = 0, n_err = 0, bad = 0;
            for (int iter = 0; iter < 100000 && got < total; ++iter) {
                const std::size_t room = std::min(in->free_space(), total - fed);
                if (room) {
                    auto span = in->write_span(room);
                    for (std::size_t i = 0; i < room; ++i) span[i] = static_cast<float>((fed + i) % 1021);
                    in->publish(room);
                    fed += room;
                }
                if (fed == total) in->producer_done = true;
                const auto r = run.work(std::numeric_limits<std::size_t>::max());
                if (r.status == work::Status::ERROR) ++n_err;
                const std::size_t avail = outb->available();
                auto rs = outb->read_span(avail);
                for (std::size_t i = 0; i < avail; ++i) bad += rs[i] != 3.f * static_cast<float>((got + i) % 1021);
                outb->consume(avail);
                got += avail;
                if (r.status == work::Status::DONE) break;
            }
            const bool ok = n_err == 1 && got == total && bad == 0 && in->available() == 0;
            std::printf("a failing launch in mid-stream (%s %s edges): %s (%zu errors, %zu of %zu samples, %zu wrong, %zu launches)\n", cap >> 20 ? "chunk-sized" : "65536-item", pinned ? "page-locked" : "ordinary",
                        ok ? "recovered" : "FAILED", n_err, got, total, bad, run.launches());
            if (!ok) ++errors;
        }
    }
    std::printf(errors ? "host-device: %d FAILURES\n" : "host-device: all graphs ran\n", errors);
    return errors ? 1 : 0;
}
