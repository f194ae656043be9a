// ref_harness.cpp -- builds oracle/_ref/libgr4ref.so FROM THE REFERENCE'S OWN HEADERS where they lie under
// /root/reference (never copied).  Only leaf headers compile with this image's toolchain without stand-ins (g++ 11 has no
// <format>/<print>/<expected>; see DESIGN.md "Oracle"): the two rng headers and the three signal-generator headers on top of them
// (algorithm/.../signal/{ToneGenerator,NoiseGenerator,SignalGeneratorCore}.hpp).  TEST INFRASTRUCTURE ONLY.
#include <complex>
#include <cstddef>
#include <cstdint>
#include <span>

#include <gnuradio-4.0/algorithm/rng/GaussianNoise.hpp>
#include <gnuradio-4.0/algorithm/rng/Xoshiro256pp.hpp>
#include <gnuradio-4.0/algorithm/signal/SignalGeneratorCore.hpp>

namespace {
template <typename T>
void signal_generate(int type, float frequency, float sample_rate, float phase, float amplitude, float offset, std::uint64_t seed, T* out, std::size_t n) {
    gr::signal::SignalGeneratorCore<T> core; // as gr::basic::SignalGenerator<T>::start() + processBulk (blocks/basic/.../SignalGenerator.hpp:59-83): configure, reset, one
    core.configure(static_cast<gr::signal::SignalType>(type), frequency, sample_rate, phase, amplitude, offset, seed); // generateSample() per output sample
    core.reset();
    for (std::size_t i = 0; i < n; ++i) out[i] = core.generateSample();
}
} // namespace

extern "C" {
void gr4ref_xoshiro_draws(std::uint64_t seed, std::uint64_t* out, std::size_t n) {
    gr::rng::Xoshiro256pp rng(seed);
    for (std::size_t i = 0; i < n; ++i) out[i] = rng();
}
void gr4ref_gauss_fill_f32(std::uint64_t seed, float* out, std::size_t n, float amplitude, float offset) {
    gr::rng::Xoshiro256pp        rng(seed);
    gr::rng::GaussianNoise<float> g(rng);
    g.fill(std::span<float>(out, n), amplitude, offset);
}
void gr4ref_gauss_fill_c32(std::uint64_t seed, float* out_interleaved, std::size_t n, float amplitude, float offset) {
    gr::rng::Xoshiro256pp        rng(seed);
    gr::rng::GaussianNoise<float> g(rng);
    g.fillComplex(std::span<std::complex<float>>(reinterpret_cast<std::complex<float>*>(out_interleaved), n), amplitude, offset);
}
void gr4ref_signal_f32(int type, float f, float fs, float ph, float a, float o, std::uint64_t seed, float* out, std::size_t n) { signal_generate<float>(type, f, fs, ph, a, o, seed, out, n); }
void gr4ref_signal_f64(int type, float f, float fs, float ph, float a, float o, std::uint64_t seed, double* out, std::size_t n) { signal_generate<double>(type, f, fs, ph, a, o, seed, out, n); }
void gr4ref_signal_i16(int type, float f, float fs, float ph, float a, float o, std::uint64_t seed, std::int16_t* out, std::size_t n) { signal_generate<std::int16_t>(type, f, fs, ph, a, o, seed, out, n); }
void gr4ref_signal_c32(int type, float f, float fs, float ph, float a, float o, std::uint64_t seed, float* out_interleaved, std::size_t n) {
    signal_generate<std::complex<float>>(type, f, fs, ph, a, o, seed, reinterpret_cast<std::complex<float>*>(out_interleaved), n);
}
}
