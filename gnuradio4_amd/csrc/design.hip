// design.hip -- host-side filter design used by the BasicFilter mirror (no device code).
//
// Restates gr::filter::fir::designFilter<float> (algorithm/.../filter/FilterTool.hpp:964-1071) and
// gr::filter::iir::designFilter<float> (:476-917; biquad sections, maxSectionSize == 2) so that a
// BasicFilterProto<float> built on this library designs the same taps / sections as the reference block
// (blocks/filter/.../time_domain_filter.hpp:163-182).  Quirks kept on purpose: bilinear
// transform without pre-warping, zeros at infinity are NOT mapped to z = -1 (all-pole low-pass sections),
// every section is gain-normalised separately, band-pass FIR = LP(f_low) - LP(f_high).
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <complex>
#include <limits>

namespace {
using cd = std::complex<double>;
using namespace gr4;

struct PZ { std::vector<cd> poles, zeros; double gain = 1.0; };

double pz_mag(const PZ& v, double omega) { // calculateResponse<RadianPerSec, Magnitude> (:459-474)
    const cd iw(0, omega);
    cd       num(1.0), den(1.0);
    for (auto& z : v.zeros) num *= (iw - z);
    for (auto& p : v.poles) den *= (iw - p);
    return v.gain * std::abs(num / den);
}

bool proto(int design, size_t order, double rippleDb, double attDb, PZ& r) {
    r = PZ{};
    switch (design) {
    case 0: { // Butterworth (:476-494)
        if (order % 2) r.poles.emplace_back(-1.0);
        for (size_t i = 0; i < order / 2; ++i) {
            const double th = M_PI * (1.0 - double(i * 2 + 1 + order % 2) / (2.0 * double(order)));
            const cd     p  = std::polar(1.0, th);
            r.poles.emplace_back(p.real(), +p.imag());
            r.poles.emplace_back(p.real(), -p.imag());
        }
        return true;
    }
    case 1: { // Bessel table (:496-513)
        static const double g[11] = {1.0, 1.0, 1.6221, 2.9067, 5.1002, 11.9773, 26.8334, 41.5419, 183.3982, 306.9539, 1893.1098};
        static const std::vector<std::vector<double>> t = {
            {-1.0, 0}, {-1.0, 0}, {-1.1030, 0.6368}, {-1.0509, 0, -1.3270, 1.0025}, {-1.3596, 0.4071, -0.9877, 1.2476},
            {-1.3851, 0, -0.9606, 1.4756, -1.5069, 0.7201}, {-1.5735, 0.3213, -1.3836, 0.9727, -0.9318, 1.6640},
            {-1.6130, 0, -1.3797, 0.5896, -1.1397, 1.1923, -0.9104, 1.8375}, {-1.7627, 0.2737, -0.8955, 2.0044, -1.3780, 0.8253, -1.6419, 1.3926},
            {-1.8081, 0, -1.6532, 0.5126, -1.16532, 1.0319, -1.3683, 1.5685, -0.8788, 2.1509},
            {-1.9335, 0.2424, -0.8684, 2.2996, -1.8478, 0.7295, -1.6669, 1.2248, -1.3649, 1.7388}};
        if (order > 10) return false;
        const auto& row = t[order];
        for (size_t i = 0; i + 1 < row.size(); i += 2) {
            if (row[i + 1] == 0) r.poles.emplace_back(row[i], 0.0);
            else { r.poles.emplace_back(row[i], row[i + 1]); r.poles.emplace_back(row[i], -row[i + 1]); }
        }
        r.gain = g[order];
        return true;
    }
    case 2: { // Chebyshev I (:515-531)
        const double eps = std::sqrt(std::pow(10, rippleDb / 10.0) - 1), sh = std::asinh(1 / eps) / double(order);
        for (size_t k = 0; k < order; ++k) {
            const double a = M_PI * (double(k) + 0.5) / double(order);
            r.poles.emplace_back(-std::sinh(sh) * std::sin(a), std::cosh(sh) * std::cos(a));
        }
        r.gain = 1.0 / pz_mag(r, 0.0);
        return true;
    }
    case 3: { // Chebyshev II (:533-562)
        const double eps = 1.0 / std::sqrt(std::pow(10, attDb / 10.0) - 1), v0 = std::asinh(1.0 / eps) / double(order);
        const double s0 = -std::sinh(v0), c0 = std::cosh(v0);
        for (size_t k = 1; k < order; k += 2) {
            const double th = 0.5 * (double(k) - double(order)) / double(order);
            const double a = s0 * std::cos(M_PI * th), b = c0 * std::sin(M_PI * th), d2 = a * a + b * b;
            r.poles.emplace_back(a / d2, +b / d2);
            r.poles.emplace_back(a / d2, -b / d2);
            const double im = 1.0 / std::cos(0.5 * M_PI * double(k) / double(order));
            r.zeros.emplace_back(0, +im);
            r.zeros.emplace_back(0, -im);
        }
        if (order & 1) r.poles.emplace_back(1.0 / s0, 0);
        r.gain = 1.0 / pz_mag(r, 0.0);
        return true;
    }
    }
    return false;
}

PZ to_response(int response, const PZ& lp, const gr4hip_filter_params& p) { // :678-819
    PZ o;
    if (response == 0) { // low-pass
        o = lp;
        for (auto& q : o.poles) q *= 2. * M_PI * p.f_low;
        for (auto& q : o.zeros) q *= 2. * M_PI * p.f_low;
        o.gain = p.gain * lp.gain / pz_mag(o, 0.);
    } else if (response == 1) { // high-pass
        o = lp;
        for (auto& q : o.poles) q = 2. * M_PI * p.f_high / q;
        if (o.zeros.empty()) o.zeros.resize(lp.poles.size());
        else {
            for (auto& q : o.zeros) q = 2. * M_PI * p.f_high / q;
            if (lp.poles.size() > o.zeros.size()) o.zeros.resize(lp.poles.size());
        }
        const double nf = std::isfinite(p.fs) ? p.fs : 10 * p.f_high;
        o.gain = p.gain * lp.gain / pz_mag(o, M_PI * nf);
    } else {
        const double w0 = 2. * M_PI * std::sqrt(p.f_low * p.f_high), bw = 2. * M_PI * std::abs(p.f_high - p.f_low), Q = w0 / bw;
        auto bp = [&](const cd& s) { const cd disc = 2.0 * w0 * std::sqrt(s * s / (4.0 * Q * Q) - 1.0), base = (w0 / Q) * s; return std::pair<cd, cd>{0.5 * (base + disc), 0.5 * (base - disc)}; };
        auto bs = [&](const cd& s) { const cd disc = 0.5 * w0 * std::sqrt(1.0 / (Q * Q * s * s) - 4.0), base = 0.5 * w0 / (Q * s); return std::pair<cd, cd>{base + disc, base - disc}; };
        for (auto& q : lp.poles) { auto [a, b] = response == 2 ? bp(q) : bs(q); o.poles.push_back(a); o.poles.push_back(b); }
        for (auto& q : lp.zeros) {
            if (std::norm(q) < 1e-10) { o.zeros.emplace_back(0., +w0); o.zeros.emplace_back(0., -w0); }
            else { auto [a, b] = response == 2 ? bp(q) : bs(q); o.zeros.push_back(a); o.zeros.push_back(b); }
        }
        if (response == 2) {
            if (lp.poles.size() > lp.zeros.size()) for (size_t i = 0; i < lp.poles.size() - lp.zeros.size(); ++i) o.zeros.emplace_back(0., 0.);
            o.gain = p.gain / pz_mag(o, w0);
        } else {
            for (size_t i = 0; i < lp.poles.size() - lp.zeros.size(); ++i) { o.zeros.emplace_back(0., +w0); o.zeros.emplace_back(0., -w0); }
            o.gain = p.gain / pz_mag(o, 0.);
        }
    }
    return o;
}

std::vector<cd> sort_conj(const std::vector<cd>& v) { // :585-624
    std::vector<cd> pos, neg, re, out;
    for (auto& c : v) (c.imag() > 1e-10 ? pos : c.imag() < -1e-10 ? neg : re).push_back(c);
    auto cmp = [](const cd& a, const cd& b) { return a.real() < b.real(); };
    std::sort(pos.begin(), pos.end(), cmp);
    std::sort(neg.begin(), neg.end(), cmp);
    std::sort(re.begin(), re.end(), cmp);
    for (size_t i = 0; i < neg.size(); ++i) { out.push_back(neg[i]); if (i < pos.size()) out.push_back(pos[i]); }
    for (size_t i = neg.size(); i < pos.size(); ++i) out.push_back(pos[i]);
    out.insert(out.end(), re.begin(), re.end());
    return out;
}

bool expand_roots(const cd* roots, size_t n, size_t desiredOrder, std::vector<float>& c) { // expandRootsToPolynomial<float> (:628-676)
    if (n == 0) { c.assign(desiredOrder + 1, 0.f); c[0] = 1.f; return true; }
    c = {1.f};
    const float eps = 1e-10f;
    for (size_t i = 0; i < n; ++i) {
        const std::complex<float> r((float)roots[i].real(), (float)roots[i].imag());
        std::vector<float>        f;
        if (std::abs(r.imag()) > eps) {
            if (i + 1 >= n) return false;
            const std::complex<float> q((float)roots[i + 1].real(), (float)roots[i + 1].imag());
            if (std::abs(r.real() - q.real()) > eps || std::abs(r.imag() + q.imag()) > eps) return false;
            f = {1.f, -(2 * r.real()), std::norm(r)};
            ++i;
        } else f = {1.f, -r.real()};
        std::vector<float> o(c.size() + f.size() - 1, 0.f);
        for (size_t a = 0; a < c.size(); ++a)
            for (size_t b = 0; b < f.size(); ++b) o[a + b] += c[a] * f[b];
        c = o;
    }
    return true;
}

float response_mag_f(const std::vector<float>& b, const std::vector<float>& a, float fnorm) { // :379-423 with T = float
    const std::complex<float> iOmega = std::polar(1.f, 2.f * 3.14159265358979323846f * fnorm);
    auto acc = [&](const std::vector<float>& c) {
        std::complex<float> s(0);
        for (size_t n = 0; n < c.size(); ++n) s += c[n] * static_cast<std::complex<float>>(std::pow(iOmega, -static_cast<int>(n)));
        return s;
    };
    return std::abs(acc(b) / acc(a));
}

void fir_generate(size_t N, int window, float fc, float beta, std::vector<float>& c) { // generateCoefficients<float> (:964-976)
    c.resize(N);
    make_window(window, c.data(), N, beta);
    const float M = float(N - 1) / 2.f, pi = 3.14159265358979323846f;
    for (size_t i = 0; i < N; ++i) {
        const float x = 2.f * fc * (float(i) - M);
        c[i] = c[i] * 2.f * fc * (x == 0.f ? 1.f : std::sin(pi * x) / (pi * x));
    }
}
} // namespace

extern "C" {

int gr4hip_filter_params_default(gr4hip_filter_params* p) {
    GR4_REQUIRE(p, "params is null");
    p->order = 4; // FilterTool.hpp:66-75
    p->f_low = p->f_high = p->fs = std::numeric_limits<double>::quiet_NaN();
    p->gain = 1.0; p->ripple_db = 0.1; p->attenuation_db = 40; p->beta = 1.6;
    return GR4HIP_OK;
}

int gr4hip_fir_design(int response, const gr4hip_filter_params* p, int window, float* h_taps, size_t cap, size_t* ntaps) {
    GR4_REQUIRE(p && ntaps, "fir_design: null argument");
    GR4_REQUIRE(response >= 0 && response <= 3, "fir_design: unknown response %d", response);
    const double minW = 0.1 / double(p->order); // estimateRequiredTransitionWidth (:993-1004)
    double       tw;
    switch (response) {
    case 0: tw = std::min(minW, std::min(std::abs(p->f_low / p->fs), std::abs(0.5 - p->f_low / p->fs))); break;
    case 1: tw = std::min(minW, std::abs(p->f_high / p->fs)); break;
    case 2: tw = std::min(minW, std::min(std::abs(p->f_low / p->fs), std::abs(0.5 - p->f_high / p->fs))); break;
    default: tw = std::min(minW, std::min(std::abs(0.5 - p->f_high / p->fs), std::min(p->f_low, 0.5 * std::abs(p->f_high - p->f_low)) / p->fs)); break;
    }
    GR4_REQUIRE(std::isfinite(tw) && tw > 0, "fir_design: invalid frequencies (fs / f_low / f_high)");
    size_t N = (size_t)std::ceil((p->attenuation_db - 8.0) / (2.285 * (2. * M_PI * tw))); // estimateNumberOfTapsKaiser (:985-991)
    if (N % 2 == 0) N++;
    *ntaps = N;
    if (!h_taps || cap < N) return h_taps ? GR4HIP_INSUFFICIENT_OUTPUT : GR4HIP_OK; // query mode
    std::vector<float> c, c2;
    float              fnorm = 0.f;
    switch (response) {
    case 0: fir_generate(N, window, (float)(p->f_low / p->fs), (float)p->beta, c); break;
    case 1:
        fir_generate(N, window, (float)(0.5 - p->f_high / p->fs), (float)p->beta, c);
        for (size_t n = 0; n < N; ++n) c[n] *= (n % 2 == 0 ? 1 : -1);
        fnorm = 0.48f;
        break;
    case 2:
        fir_generate(N, window, (float)(p->f_low / p->fs), (float)p->beta, c);
        fir_generate(N, window, (float)(p->f_high / p->fs), (float)p->beta, c2);
        for (size_t n = 0; n < N; ++n) c[n] -= c2[n];
        fnorm = (float)(std::sqrt(p->f_high * p->f_low) / p->fs);
        break;
    default:
        fir_generate(N, window, (float)(p->f_low / p->fs), (float)p->beta, c);
        fir_generate(N, window, (float)(p->f_high / p->fs), (float)p->beta, c2);
        for (size_t n = 0; n < N; ++n) { c[n] -= c2[n]; if (N % 2 != 0 && n == (N - 1) / 2) c[n] = 1 - c[n]; }
        break;
    }
    const float mag = response_mag_f(c, {1.f}, fnorm);
    if (mag == 0) { set_error("fir_design: gain correction failed (zero response at the reference frequency)"); return GR4HIP_ERROR; } // reference throws
    for (size_t n = 0; n < N; ++n) h_taps[n] = c[n] * (float)p->gain / mag;
    return GR4HIP_OK;
}

int gr4hip_iir_design(int response, const gr4hip_filter_params* p, int design, float* h_b, float* h_a, size_t cap_sections, size_t* nsections) {
    GR4_REQUIRE(p && nsections && h_b && h_a, "iir_design: null argument");
    GR4_REQUIRE(response >= 0 && response <= 3, "iir_design: unknown response %d", response);
    GR4_REQUIRE(response == 1 || std::isfinite(p->f_low), "FilterParameters::fLow is NaN -> please set");   // :834-836
    GR4_REQUIRE(response == 0 || std::isfinite(p->f_high), "FilterParameters::fHigh is NaN -> please set"); // :837-839
    GR4_REQUIRE(std::isfinite(p->fs), "FilterParameters::fs is NaN -> please set");                         // :860-862
    PZ lp;
    if (!proto(design, p->order, p->ripple_db, p->attenuation_db, lp)) { set_error("iir_design: unsupported design %d / order %zu", design, (size_t)p->order); return GR4HIP_INVALID_ARGUMENT; }
    PZ           an = to_response(response, lp, *p);
    const double twoFs = 2. * p->fs;
    for (auto& q : an.poles) q = (twoFs + q) / (twoFs - q); // bilinear, no pre-warping (:564-583)
    for (auto& q : an.zeros) q = (twoFs + q) / (twoFs - q);
    auto  poles = sort_conj(an.poles), zeros = sort_conj(an.zeros);
    float ref;
    switch (response) {
    case 2: ref = std::sqrt((float)(p->f_low * p->f_high)); break;
    case 1: ref = (float)(0.49 * p->fs); break;
    default: ref = 0.f; break;
    }
    size_t ns = 0, pi_ = 0, zi = 0;
    while (pi_ < poles.size()) { // :883-916 with maxSectionSize == 2
        const size_t np = ((float)std::abs(poles[pi_].imag()) > (float)1e-10) ? std::min<size_t>(2, poles.size() - pi_) : 1;
        const size_t nz = std::min<size_t>(2, zeros.size() - zi);
        std::vector<float> a, b;
        if (!expand_roots(poles.data() + pi_, np, np, a) || !expand_roots(zeros.data() + zi, nz, np, b)) { set_error("iir_design: roots are not conjugate pairs"); return GR4HIP_ERROR; }
        pi_ += np;
        zi += nz;
        const float mag = response_mag_f(b, a, ref / (float)p->fs);
        if (mag == 0) { set_error("iir_design: biquad gain correction failed"); return GR4HIP_ERROR; }
        if (ns >= cap_sections) return GR4HIP_INSUFFICIENT_OUTPUT;
        for (size_t j = 0; j < 3; ++j) {
            h_b[ns * 3 + j] = j < b.size() ? b[j] * (float)p->gain / mag : 0.f;
            h_a[ns * 3 + j] = j < a.size() ? a[j] : 0.f;
        }
        ++ns;
    }
    *nsections = ns;
    return GR4HIP_OK;
}

} // extern "C"
