// The sharded 8-channel graph of BASELINE.json configs[4] through the host engine (gr4/hip.hpp: plan_sharded, FanInRun):
//   source_c -> fir_filter<complex<float>> -> PowerSpectrum -> Add<float>.in[c]  (compute_domain "gpu:hip:{c mod N}")  -> sink
//   test_host_fanin <in_c32.bin> <taps.bin> <fftSize> <out_prefix> [channels = 8]
// Channel c reads the input stream rotated by 977 c samples.  With N = 1 (the one-GPU box) every channel maps to gpu:hip:0 and the communicator has ONE rank:
// the RCCL calls execute (ncclCommInitRank, ncclAllReduce / ncclReduceScatter on the run's stream), only the transport does not.  The rank-0-of-2 plan is
// checked without running it (its partner would sit on gpu:hip:1).  Exit code 0: everything ran; 3: a device call failed; 4: no RCCL library.
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

struct Built {
    Graph                          g;
    testing::VectorSink<float>*    sink = nullptr;
};
// the graph as a user writes it: n branches, branch c placed on device c mod n_devices
static void build(Built& b, const std::vector<std::complex<float>>& x, const std::vector<double>& taps, std::size_t N, std::size_t channels, int n_devices, bool distinct_taps) {
    auto& add = b.g.emplaceBlock<blocks::math::Add<float>>(property_map{{"n_inputs", std::int64_t(channels)}});
    for (std::size_t c = 0; c < channels; ++c) {
        const std::string dom = "gpu:hip:" + std::to_string(static_cast<int>(c) % n_devices);
        auto& src  = b.g.emplaceBlock<testing::VectorSource<std::complex<float>>>();
        if (c == 1) src._tags.push_back(Tag{5 * N + 17, property_map{{"gr:trigger_name", "burst"s}, {"private_key", 1.f}}}); // inside the second exchange of four frames
        if (c == 2) src._tags.push_back(Tag{0, property_map{{"gr:sample_rate", 2.0e6f}}});
        src.values.resize(x.size());
        for (std::size_t i = 0; i < x.size(); ++i) src.values[i] = x[(i + 977 * c) % x.size()];
        std::vector<double> t = taps;
        if (distinct_taps)
            for (auto& v : t) v *= 1.0 + 0.125 * static_cast<double>(c); // channel c's own gain: the run cannot share H between the channels
        auto& fir  = b.g.emplaceBlock<filter::fir_filter<std::complex<float>>>(property_map{{"b", t}, {"compute_domain", dom}});
        auto& spec = b.g.emplaceBlock<blocks::fft::PowerSpectrum<std::complex<float>>>(property_map{{"fftSize", std::int64_t(N)}, {"window", "None"s}, {"compute_domain", dom}});
        if (!b.g.connect<"out", "in">(src, fir) || !b.g.connect<"out", "in">(fir, spec) || !b.g.connect(spec, "out"s, add, "in#"s + std::to_string(c))) { std::fprintf(stderr, "connect failed\n"); std::exit(2); }
    }
    b.sink = &b.g.emplaceBlock<testing::VectorSink<float>>();
    if (!b.g.connect<"out", "in">(add, *b.sink)) { std::fprintf(stderr, "connect failed\n"); std::exit(2); }
}

int main(int argc, char** argv) {
    if (argc < 5) { std::fprintf(stderr, "usage: %s in_c32.bin taps.bin fftSize out_prefix [channels]\n", argv[0]); return 2; }
    const auto                x    = load<std::complex<float>>(argv[1]);
    const auto                taps = load<float>(argv[2]);
    const std::size_t         N    = std::stoul(argv[3]);
    const std::string         out  = argv[4];
    const std::size_t         C    = argc > 5 ? std::stoul(argv[5]) : 8;
    const std::vector<double> tapsd(taps.begin(), taps.end());
    int errors = 0;

    // ---- the communicator: one rank, over librccl
    char id[GR4HIP_FANIN_ID_BYTES];
    if (const int rc = gr4hip_fanin_unique_id(id); rc != 0) { std::fprintf(stderr, "fan-in unavailable: %s\n", gr4hip_last_error()); return rc == GR4HIP_UNSUPPORTED ? 4 : 3; }
    gr4hip_fanin_t* comm = nullptr;
    if (gr4hip_set_device(0) != 0 || gr4hip_fanin_create(&comm, id, 0, 1) != 0) { std::fprintf(stderr, "communicator: %s\n", gr4hip_last_error()); return 3; }
    int rank = -1, n_ranks = -1;
    gr4hip_fanin_rank(comm, &rank, &n_ranks);
    std::printf("communicator: rank %d of %d\n", rank, n_ranks);

    for (int variant = 0; variant < 3; ++variant) { // 0: shared taps, all_reduce; 1: own taps per channel, all_reduce; 2: shared taps, reduce_scatter (one rank: its shard is everything)
        Built b;
        build(b, x, tapsd, N, C, 1, variant == 1);
        const std::size_t before = b.g.blocks().size();
        hip::Shard shard{0, 1, comm, variant == 2, 4};
        auto runs = hip::plan_sharded(b.g, shard);
        if (runs.size() != 1 || runs[0]->local_channels() != C) { std::printf("plan (variant %d): FAILED (%zu runs)\n", variant, runs.size()); return 3; }
        std::printf("plan (variant %d): %zu blocks -> %zu; %s\n", variant, before, b.g.blocks().size(), std::string(runs[0]->name()).c_str());
        if (b.g.blocks().size() != C + 2) ++errors; // the sources, the run, the sink
        scheduler::Simple sched;
        auto* run = runs[0];
        sched.exchange(std::move(b.g));
        if (const auto r = sched.runAndWait(); !r) { std::cerr << "sharded graph: " << r.error().message << "\n"; return 3; }
        const std::size_t frames = x.size() / N, want_launches = (frames + 3) / 4;
        std::printf("run (variant %d): %zu outputs, %zu launches (%zu queued beside an exchange still in flight), %zu collectives\n", variant, b.sink->_samples.size(), run->launches(), run->overlapped(), run->exchanges());
        if (b.sink->_samples.size() != frames * N || run->launches() != want_launches || run->exchanges() != want_launches) ++errors;
        if (want_launches > 2 && run->overlapped() == 0) { std::printf("run (variant %d): no launch was ever queued beside the exchange before it\n", variant); ++errors; } // the two-slab pipeline
        { // tags cross the run like one n-ary block: channel 2's rate tag on sample 0, channel 1's trigger at the start of the exchange it fell into, "gr:" keys only
            const auto& tg = b.sink->_tags;
            const bool ok = C < 3 || (run->tags_forwarded() == 2 && tg.size() == 2 && tg[0].index == 0 && tg[0].map.count("gr:sample_rate") && std::get<float>(tg[0].map.at("gr:sample_rate")) == 2.0e6f &&
                                      tg[1].index == 4 * N && tg[1].map.count("gr:trigger_name") && !tg[1].map.count("private_key"));
            std::printf("tags (variant %d): %s (%zu forwarded, %zu received)\n", variant, ok ? "ok" : "FAILED", run->tags_forwarded(), tg.size());
            if (!ok) ++errors;
        }
        dump(out + "_fanin" + std::to_string(variant) + ".bin", b.sink->_samples);
    }
    { // the same graph placed on TWO devices, planned for rank 0 (not run: its partner would sit on gpu:hip:1): half of the branches stay, the others leave with their sources
        Built b;
        build(b, x, tapsd, N, C, 2, false);
        hip::Shard shard{0, 2, nullptr, true, 4};
        auto runs = hip::plan_sharded(b.g, shard);
        const bool ok = runs.size() == 1 && runs[0]->local_channels() == (C + 1) / 2 && b.g.blocks().size() == (C + 1) / 2 + 2;
        std::printf("plan for rank 0 of 2: %s (%zu local channels, %zu blocks)\n", ok ? "ok" : "FAILED", runs.empty() ? std::size_t(0) : runs[0]->local_channels(), b.g.blocks().size());
        if (!ok) ++errors;
    }
    { // a rank that owns no branch could never take part in the exchanges: planning refuses instead of leaving its partners waiting in a collective
        Built b;
        build(b, x, tapsd, N, C, 1, false);
        bool refused = false;
        try { (void)hip::plan_sharded(b.g, hip::Shard{1, 2, nullptr, false, 4}); } catch (const std::exception& e) { refused = std::string(e.what()).find("owns no branch") != std::string::npos; }
        std::printf("plan for a rank without branches: %s\n", refused ? "refused" : "FAILED");
        if (!refused) ++errors;
    }
    gr4hip_fanin_destroy(comm);
    std::printf(errors ? "FAILED (%d)\n" : "all sharded-graph checks passed\n", errors);
    return errors ? 3 : 0;
}
