// gr::basic::SignalGenerator<T> of the host mirror through Graph + scheduler, every signal type, four output types -> raw files that
// tests/test_host_cpp.py compares with the reference's own SignalGeneratorCore<T> (oracle/_ref, built from the reference's headers where they lie).
//   dump_signal_generator <out_prefix> <n_samples>
#include <cstdio>
#include <fstream>

#include <gr4/blocks.hpp>

using namespace gr;
using namespace std::string_literals;

template <typename T>
int dump(const std::string& prefix, const char* tname, std::size_t n, float amplitude, float offset) {
    static constexpr const char* names[] = {"Const", "Sin", "Cos", "Square", "Saw", "Triangle", "FastSin", "FastCos", "UniformNoise", "TriangularNoise", "GaussianNoise"};
    int bad = 0;
    for (int type = 0; type < 11; ++type) {
        Graph g;
        auto& src  = g.emplaceBlock<basic::SignalGenerator<T>>({{"signal_type", std::string(names[type])}, {"frequency", 37.5}, {"sample_rate", 1000.0}, {"phase", 0.3}, {"amplitude", double(amplitude)},
                                                                {"offset", double(offset)}, {"seed", std::int64_t(12345)}, {"n_samples_max", std::int64_t(n)}});
        auto& sink = g.emplaceBlock<testing::VectorSink<T>>();
        if (!g.connect<"out", "in">(src, sink)) return 1;
        scheduler::Simple sched;
        sched.exchange(std::move(g));
        if (!sched.runAndWait() || sink._samples.size() != n) ++bad;
        std::ofstream f(prefix + "_" + tname + "_" + std::to_string(type) + ".bin", std::ios::binary);
        f.write(reinterpret_cast<const char*>(sink._samples.data()), static_cast<std::streamsize>(sink._samples.size() * sizeof(T)));
    }
    return bad;
}

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s out_prefix n_samples\n", argv[0]); return 2; }
    const std::string prefix = argv[1];
    const std::size_t n      = std::stoul(argv[2]);
    int bad = dump<float>(prefix, "f32", n, 1.5f, 0.25f) + dump<double>(prefix, "f64", n, 1.5f, 0.25f) + dump<std::int16_t>(prefix, "i16", n, 30000.f, 9000.f) +
              dump<std::complex<float>>(prefix, "c32", n, 1.5f, 0.25f);
    std::printf(bad ? "signal generator dump: %d FAILURES\n" : "signal generator dump: ok\n", bad);
    return bad ? 1 : 0;
}
