// ref_harness.cpp -- builds oracle/_ref/libgr4ref.so FROM THE REFERENCE'S OWN HEADERS where they lie under
// /root/reference (never copied).  Only the two rng leaf headers compile with this image's toolchain without
// stand-ins (g++ 11 has no <format>/<print>/<expected>; see DESIGN.md "Oracle").  TEST INFRASTRUCTURE ONLY.
#include <complex>
#include <cstddef>
#include <cstdint>
#include <span>

#include <gnuradio-4.0/algorithm/rng/GaussianNoise.hpp>
#include <gnuradio-4.0/algorithm/rng/Xoshiro256pp.hpp>

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
}
