// The plugin entry (Plugin.hpp:82-85) end to end:   test_host_plugin <libgr4hip_blocks.so> <compute_domain> [not_a_plugin.so]
// dlopen -> gr_plugin_make -> ABI check -> blocks by registry name from property_maps -> Graph::addBlock / connect by port name -> run.
// This program does NOT link the plugin or libgr4hip.so: everything it runs comes through the two C entry points.
// compute_domain "host": the blocks' host bodies (works on any box); "gpu:hip:0": the device seam (fails loudly, exit code 3, without a GPU).
#include <cstdio>
#include <iostream>

#include <gr4/blocks.hpp>
#include <gr4/grc.hpp>
#include <gr4/plugin.hpp>

using namespace gr;
using namespace std::string_literals;

static int failures = 0;
#define EXPECT(cond) do { if (!(cond)) { ++failures; std::printf("FAIL %s:%d  %s\n", __FILE__, __LINE__, #cond); } } while (0)

int main(int argc, char** argv) {
    if (argc < 3) { std::fprintf(stderr, "usage: %s plugin.so compute_domain [other.so]\n", argv[0]); return 2; }
    const std::string domain = argv[2];
    PluginLoader loader;
    if (argc > 3) { // a library without the entry points is refused with a reason, not a crash
        const auto r = loader.load(argv[3]);
        EXPECT(!r.has_value() && loader.failedPlugins().size() == 1);
        if (!r) std::printf("refused: %s\n", r.error().message.c_str());
    }
    EXPECT(!loader.load("/nonexistent/libnope.so").has_value());
    const auto ok = loader.load(argv[1]);
    if (!ok) { std::fprintf(stderr, "%s\n", ok.error().message.c_str()); return 2; }
    const auto names = loader.availableBlocks();
    std::printf("plugin blocks: %zu\n", names.size());
    for (const char* n : {"gr::filter::fir_filter<float32>", "gr::filter::fir_filter<complex<float32>>", "gr::filter::iir_filter<float32, gr::filter::IIRForm::DF_II>",
                          "gr::filter::BasicFilter<float32>", "gr::filter::BasicFilterProto<float32, gr::Resampling<1, 1, false>>", "gr::filter::Decimator<complex<float64>>",
                          "gr::blocks::math::AddConst<uint8>", "gr::blocks::math::Divide<int64>", "gr::blocks::math::Rotator<complex<float32>>", "gr::blocks::fft::FFT<float32>",
                          "gr::blocks::fft::FFT<complex<float32>>"})
        EXPECT(loader.isBlockAvailable(n));
    EXPECT(loader.instantiate("gr::no::such_block<float32>") == nullptr);
    bool threw = false;
    try { (void)loader.instantiate("gr::filter::fir_filter<float32>", {{"no_such_setting", 1.0}}); } catch (const std::invalid_argument&) { threw = true; }
    EXPECT(threw);

    // graph purely from the registry: source -> MultiplyConst<int32> (x3, wraps) -> sink, and source -> fir_filter<float32> -> Decimator<float32> -> sink
    Graph g;
    auto& isrc  = g.addBlock(loader.instantiate("gr::testing::VectorSource<int32>", {{"n_samples_max", std::int64_t(100000)}}));
    auto& imul  = g.addBlock(loader.instantiate("gr::blocks::math::MultiplyConst<int32>", {{"value", std::int64_t(3)}, {"compute_domain", domain}}));
    auto& isink = g.addBlock(loader.instantiate("gr::testing::VectorSink<int32>"));
    static_cast<testing::VectorSource<std::int32_t>*>(isrc.raw())->values = {2147483647, -5, 7, 123456789};
    EXPECT(g.connect(isrc, "out"s, imul, "in"s).has_value() && g.connect(imul, "out"s, isink, "in"s).has_value());
    EXPECT(!g.connect(isrc, "out"s, isink, "nope"s).has_value());
    auto& fsrc  = g.addBlock(loader.instantiate("gr::testing::VectorSource<float32>", {{"n_samples_max", std::int64_t(200000)}}));
    auto& ffir  = g.addBlock(loader.instantiate("gr::filter::fir_filter<float32>", {{"b", std::vector<double>{0.5, 0.25, 0.25}}, {"compute_domain", domain}}));
    auto& fdec  = g.addBlock(loader.instantiate("gr::filter::Decimator<float32>", {{"decim", std::int64_t(4)}, {"compute_domain", domain}}));
    auto& fsink = g.addBlock(loader.instantiate("gr::testing::VectorSink<float32>"));
    static_cast<testing::VectorSource<float>*>(fsrc.raw())->values = {1.f, -2.f, 3.f, 0.5f, 0.25f};
    static_cast<testing::VectorSource<float>*>(fsrc.raw())->_tags  = {{0, {{"gr:sample_rate", 48000.f}}}};
    EXPECT(g.connect(fsrc, "out"s, ffir, "in"s).has_value() && g.connect(ffir, "out"s, fdec, "in"s).has_value() && g.connect(fdec, "out"s, fsink, "in"s).has_value());
    EXPECT(ffir.type_name() == "gr::filter::fir_filter<float32>" && ffir.compute_domain().is_device() == (domain != "host"));

    auto sched = loader.instantiateScheduler("gr::scheduler::Simple");
    EXPECT(sched != nullptr && loader.instantiateScheduler("gr::scheduler::Nope") == nullptr);
    if (!sched) return 1;
    sched->exchange(std::move(g));
    if (const auto r = sched->runAndWait(); !r) {
        std::fprintf(stderr, "plugin graph: %s\n", r.error().message.c_str());
        return 3; // a device domain without a device: ERROR from the seam, never a host fallback
    }
    const auto& iv = static_cast<testing::VectorSink<std::int32_t>*>(isink.raw())->_samples;
    const auto& fv = static_cast<testing::VectorSink<float>*>(fsink.raw())->_samples;
    const auto& ft = static_cast<testing::VectorSink<float>*>(fsink.raw())->_tags;
    EXPECT(iv.size() == 100000u && fv.size() == 50000u);
    const std::int32_t pat[] = {2147483647, -5, 7, 123456789};
    bool iok = true, fok = true;
    for (std::size_t i = 0; i < iv.size() && iok; ++i) iok = iv[i] == static_cast<std::int32_t>(static_cast<std::int64_t>(pat[i % 4]) * 3);
    const float fpat[] = {1.f, -2.f, 3.f, 0.5f, 0.25f};
    for (std::size_t m = 0; m < fv.size() && fok; ++m) {
        const std::size_t n = 4 * m;
        const double want = 0.5 * fpat[n % 5] + (n >= 1 ? 0.25 * fpat[(n - 1) % 5] : 0.0) + (n >= 2 ? 0.25 * fpat[(n - 2) % 5] : 0.0);
        fok = std::abs(fv[m] - want) <= 1e-5;
    }
    EXPECT(iok && fok);
    EXPECT(ft.size() == 1u && ft[0].index == 0u && (ft[0].map == property_map{{"gr:sample_rate", 12000.f}})); // through fir (1:1) and Decimator (/4)
    { // a graph description in the GRC / YAML format (shape of core/test/qa_grc.cpp:132-152): ids from the plugin's registry, compute_domain per block,
      // connections by port index, [index, sub-index] and name
        const std::string yaml = R"(
blocks:
  - id: gr::testing::VectorSource<float32>
    parameters:
      name: ramp                      # comments and blank lines are fine

      values: [1, -2, 3, 0.5, 0.25]
      n_samples_max: 40000
  - id: gr::testing::VectorSource<float32>
    parameters:
      name: ones
      values: [1.0]
      n_samples_max: !!uint32 40000
  - id: gr::blocks::math::Add<float32>
    parameters:
      name: sum
      n_inputs: 2
      compute_domain: "DOMAIN"
  - id: gr::filter::fir_filter<float32>
    parameters:
      name: 'smooth'
      b: [0.5, 0.25, 0.25]
      compute_domain: "DOMAIN"
  - id: gr::filter::Decimator<float32>
    parameters: 
      name: every4th
      decim: 4
      compute_domain: "DOMAIN"
  - id: gr::testing::VectorSink<float32>
    parameters:
      name: sink
connections:
  - [ramp, 0, sum, [0, 0]]
  - [ones, [0, 0], sum, [0, 1]]
  - [sum, out, smooth, in]
  - [smooth, 0, every4th, 0]
  - [every4th, 0, 'sink', 0]
)";
        std::string text = yaml;
        for (std::size_t at; (at = text.find("DOMAIN")) != std::string::npos;) text.replace(at, 6, domain);
        Graph gg;
        auto  loaded = loadGrc(loader, gg, text);
        if (!loaded) std::printf("grc: %s\n", loaded.error().message.c_str());
        EXPECT(loaded.has_value() && loaded.value().size() == 6u);
        if (loaded) {
            auto sched2 = loader.instantiateScheduler("gr::scheduler::Simple");
            sched2->exchange(std::move(gg));
            if (const auto r = sched2->runAndWait(); !r) { std::fprintf(stderr, "grc graph: %s\n", r.error().message.c_str()); return 3; }
            const auto& y = static_cast<testing::VectorSink<float>*>(loaded.value()[5].model->raw())->_samples;
            const float pat2[] = {1.f, -2.f, 3.f, 0.5f, 0.25f};
            bool        ok     = y.size() == 10000u;
            for (std::size_t m = 0; ok && m < y.size(); ++m) {
                const std::size_t n = 4 * m;
                const auto        u = [&](std::size_t i) { return double(pat2[i % 5]) + 1.0; };
                ok = std::abs(y[m] - (0.5 * u(n) + (n >= 1 ? 0.25 * u(n - 1) : 0.0) + (n >= 2 ? 0.25 * u(n - 2) : 0.0))) <= 1e-5;
            }
            EXPECT(ok);
            EXPECT(loaded.value()[3].model->compute_domain().is_device() == (domain != "host"));
        }
        Graph bad;
        EXPECT(!loadGrc(loader, bad, "blocks:\n  - id: gr::nope<float32>\n").has_value());
        Graph bad2;
        const auto e2 = loadGrc(loader, bad2, "blocks:\n  - id: gr::testing::NullSink<float32>\n    parameters:\n      name: a\nconnections:\n  - [a, 0, b, 0]\n");
        EXPECT(!e2.has_value() && e2.error().message.find("unknown block") != std::string::npos);
    }
    if (failures) std::printf("host-plugin: %d FAILURES\n", failures);
    else std::printf("host-plugin: all checks passed (compute_domain %s)\n", domain.c_str());
    return failures ? 1 : 0;
}
