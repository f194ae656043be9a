// The reference's own algorithm/fourier/{window,fft_common}.hpp -- #included where they lie under /root/reference, UNMODIFIED -- compiled against this host layer
// (gr4/compat.hpp supplies what they name from the reference's meta/ layer: gr::meta::fixed_string, gr::meta::array_or_vector_type) and run on what the
// reference's qa_algorithm_fourier.cpp:145-180 holds: every window at N = 8 and N = 0 (create<float>, the in-place create on a std::array<double>), the numpy.unwrap vector, magnitude / phase of a small
// spectrum.  Prints the numbers; tests/test_host_cpp.py compares them with tests/golden/reference_vectors.json and with the oracle.  Nothing of the reference's
// text is kept in this repository: this program only #includes it (container-only test: skipped where /root/reference does not exist).
#include <complex>
#include <cstdio>
#include <vector>

#include <gr4/compat.hpp>

#include <gnuradio-4.0/algorithm/fourier/fft_common.hpp>
#include <gnuradio-4.0/algorithm/fourier/window.hpp>

int main() {
    namespace win = gr::algorithm::window;
    for (int t = 0; t < 12; ++t) { // window.hpp: every enumerator, float and double (create<T>(type, n) and the in-place create(container, type))
        const auto type = static_cast<win::Type>(t);
        const auto w    = win::create<float>(type, 8);
        std::printf("window %d", t);
        for (const float v : w) std::printf(" %.9g", static_cast<double>(v));
        std::printf("\n");
        std::array<double, 8> wd{};
        win::create(wd, type);
        for (std::size_t i = 0; i < 8; ++i)
            if (std::abs(wd[i] - static_cast<double>(w[i])) > 1e-6) { std::printf("FAILED: create<double> vs create<float>, window %d\n", t); return 1; }
        if (!win::create<float>(type, 0).empty()) { std::printf("FAILED: zero length\n"); return 1; }
    }
    std::printf("typenames %s\n", std::string(win::TypeNames.view()).c_str());
    std::vector<double> phase{0.2, -1.0, 2.5, -3.1, 0.9, -0.5, 1.2, 0.8, 1.5, -1.2, -2.7, 0.9, -0.8, -1.4, 0.6, 1.1, -1.9, 0.4, 1.3, -0.7};
    gr::algorithm::fft::unwrapPhase(phase);
    std::printf("unwrap");
    for (const double v : phase) std::printf(" %.9g", v);
    std::printf("\n");
    std::vector<std::complex<float>> spec{{1.f, 0.f}, {0.f, -2.f}, {-3.f, 3.f}, {0.5f, 0.25f}, {0.f, 0.f}, {2.f, -1.f}, {-1.f, -1.f}, {4.f, 0.f}};
    const auto mag  = gr::algorithm::fft::computeMagnitudeSpectrum(spec, {}, gr::algorithm::fft::ConfigMagnitude{.computeHalfSpectrum = false, .outputInDb = false, .shiftSpectrum = true});
    const auto magd = gr::algorithm::fft::computeMagnitudeSpectrum(spec, {}, gr::algorithm::fft::ConfigMagnitude{.computeHalfSpectrum = true, .outputInDb = true, .shiftSpectrum = false});
    const auto ph   = gr::algorithm::fft::computePhaseSpectrum(spec, {}, gr::algorithm::fft::ConfigPhase{.computeHalfSpectrum = false, .outputInDeg = true, .unwrapPhase = true, .shiftSpectrum = true});
    std::printf("magnitude_shifted");
    for (const float v : mag) std::printf(" %.9g", static_cast<double>(v));
    std::printf("\nmagnitude_half_db");
    for (const float v : magd) std::printf(" %.9g", static_cast<double>(v));
    std::printf("\nphase_deg_unwrapped_shifted");
    for (const float v : ph) std::printf(" %.9g", static_cast<double>(v));
    std::printf("\nreference algorithm drop-in: done (window.hpp, fft_common.hpp unmodified)\n");
    return 0;
}
