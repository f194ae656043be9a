/*
 * gr4_oracle.c -- CPU ORACLE (test infrastructure only; see gr4_oracle.h header for the rules and pinning status).
 * Plain-C restatement of the reference's hot-path arithmetic; every function cites the reference file:line it follows
 * (paths relative to /root/reference).
 */
#define _GNU_SOURCE
#include "gr4_oracle.h"

#include <complex.h>
#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* ------------------------------------------------------------------------------------------------
 * a15: Xoshiro256pp (algorithm/include/gnuradio-4.0/algorithm/rng/Xoshiro256pp.hpp:22-96)
 * ---------------------------------------------------------------------------------------------- */
static uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }

static uint64_t splitmix64(uint64_t* x) { /* Xoshiro256pp.hpp:88-93 */
    uint64_t z = (*x += 0x9e3779b97f4a7c15ULL);
    z          = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z          = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}

void gr4o_xoshiro_seed(gr4o_xoshiro_t* r, uint64_t seed) { /* :33-39 */
    uint64_t sm = seed;
    r->s[0]     = splitmix64(&sm);
    r->s[1]     = splitmix64(&sm);
    r->s[2]     = splitmix64(&sm);
    r->s[3]     = splitmix64(&sm);
}

uint64_t gr4o_xoshiro_next(gr4o_xoshiro_t* r) { /* :41-51 */
    uint64_t*      s      = r->s;
    const uint64_t result = rotl64(s[0] + s[3], 23) + s[0];
    const uint64_t t      = s[1] << 17;
    s[2] ^= s[0];
    s[3] ^= s[1];
    s[1] ^= s[2];
    s[0] ^= s[3];
    s[2] ^= t;
    s[3] = rotl64(s[3], 45);
    return result;
}

static float uniform_m11_f32(uint64_t raw) { /* toUniform01<float> :56-61, toUniformM11 :64-67 */
    return 2.0f * ((float)(raw >> 40) * 0x1.0p-24f) - 1.0f;
}

static void polar_pair_f32(gr4o_xoshiro_t* r, float* g1, float* g2) { /* GaussianNoise.hpp:101-111 */
    float u, v, s;
    do {
        u = uniform_m11_f32(gr4o_xoshiro_next(r));
        v = uniform_m11_f32(gr4o_xoshiro_next(r));
        s = u * u + v * v;
    } while (s >= 1.0f || s == 0.0f);
    const float factor = sqrtf(-2.0f * logf(s) / s);
    *g1                = u * factor;
    *g2                = v * factor;
}

void gr4o_gauss_fill_f32(gr4o_xoshiro_t* r, float* out, size_t n, float amplitude, float offset) { /* GaussianNoise.hpp:59-83 */
    int    hasSpare = 0;
    float  spare    = 0.f;
    size_t i;
    for (i = 0; i < n; ++i) {
        if (hasSpare) {
            hasSpare = 0;
            out[i]   = amplitude * spare + offset;
            continue;
        }
        float g1, g2;
        polar_pair_f32(r, &g1, &g2);
        spare    = g2;
        hasSpare = 1;
        out[i]   = amplitude * g1 + offset;
    }
}

void gr4o_gauss_fill_c32(gr4o_xoshiro_t* r, float* out, size_t n, float amplitude, float offset) { /* GaussianNoise.hpp:85-99 */
    const float scale     = 1.0f / 1.41421356237309504880f; /* F(1)/sqrt2_v<F> */
    const float scaledAmp = amplitude * scale;
    size_t      i;
    for (i = 0; i < n; ++i) {
        float g1, g2;
        polar_pair_f32(r, &g1, &g2);
        out[2 * i]     = scaledAmp * g1 + offset;
        out[2 * i + 1] = scaledAmp * g2;
    }
}

/* SURVEY.md 8(d) synthetic stream: noise (reference recipe, seed) + tone exp(2 pi i f n) computed in double */
void gr4o_signal_c32(uint64_t seed, float* out, size_t n, double tone_frel, double tone_amp, float noise_amp) {
    gr4o_xoshiro_t r;
    size_t         i;
    gr4o_xoshiro_seed(&r, seed);
    gr4o_gauss_fill_c32(&r, out, n, noise_amp, 0.f);
    for (i = 0; i < n; ++i) {
        double ph = fmod(tone_frel * (double)i, 1.0) * 2.0 * M_PI;
        out[2 * i] += (float)(tone_amp * cos(ph));
        out[2 * i + 1] += (float)(tone_amp * sin(ph));
    }
}

void gr4o_signal_f32(uint64_t seed, float* out, size_t n, double tone_frel, double tone_amp, float noise_amp) {
    gr4o_xoshiro_t r;
    size_t         i;
    gr4o_xoshiro_seed(&r, seed);
    gr4o_gauss_fill_f32(&r, out, n, noise_amp, 0.f);
    for (i = 0; i < n; ++i) out[i] += (float)(tone_amp * sin(fmod(tone_frel * (double)i, 1.0) * 2.0 * M_PI));
}

/* ------------------------------------------------------------------------------------------------
 * a1/a2: fir_filter<T>::processOne == HistoryBuffer::push_front + transform_reduce(b, history.cbegin())
 * (blocks/filter/.../time_domain_filter.hpp:44-47, core/.../HistoryBuffer.hpp:130-139, 346-363).
 * history.cbegin() is newest-first, so y[n] = sum_k b[k]*x[n-k].  `hist` = previous ntaps-1 inputs, oldest first.
 * ---------------------------------------------------------------------------------------------- */
static inline float fir_x_f32(const float* hist, const float* x, size_t ntaps, ptrdiff_t idx) {
    return idx >= 0 ? x[idx] : hist[(ptrdiff_t)(ntaps - 1) + idx];
}

static void fir_update_hist(float* hist, const float* x, size_t ntaps, size_t n, size_t ncomp) {
    const size_t H = ntaps - 1;
    size_t       i, c;
    if (H == 0) return;
    if (n >= H) {
        memcpy(hist, x + (n - H) * ncomp, H * ncomp * sizeof(float));
    } else {
        memmove(hist, hist + n * ncomp, (H - n) * ncomp * sizeof(float));
        for (i = 0; i < n; ++i)
            for (c = 0; c < ncomp; ++c) hist[(H - n + i) * ncomp + c] = x[i * ncomp + c];
    }
}

void gr4o_fir_f32(const float* b, size_t ntaps, float* hist, const float* x, float* y, size_t n) {
    size_t i, k;
    for (i = 0; i < n; ++i) {
        float acc = 0.f;
        for (k = 0; k < ntaps; ++k) acc += b[k] * fir_x_f32(hist, x, ntaps, (ptrdiff_t)i - (ptrdiff_t)k);
        y[i] = acc;
    }
    fir_update_hist(hist, x, ntaps, n, 1);
}

void gr4o_fir_f32_acc64(const float* b, size_t ntaps, float* hist, const float* x, double* y, size_t n) {
    size_t i, k;
    for (i = 0; i < n; ++i) {
        double acc = 0.;
        for (k = 0; k < ntaps; ++k) acc += (double)b[k] * (double)fir_x_f32(hist, x, ntaps, (ptrdiff_t)i - (ptrdiff_t)k);
        y[i] = acc;
    }
    fir_update_hist(hist, x, ntaps, n, 1);
}

/* complex<float> data x real float taps: same formula per component */
void gr4o_fir_c32(const float* b, size_t ntaps, float* hist, const float* x, float* y, size_t n) {
    const ptrdiff_t H = (ptrdiff_t)ntaps - 1;
    size_t          i, k;
    for (i = 0; i < n; ++i) {
        float ar = 0.f, ai = 0.f;
        for (k = 0; k < ntaps; ++k) {
            const ptrdiff_t idx = (ptrdiff_t)i - (ptrdiff_t)k;
            const float*    p   = idx >= 0 ? x + 2 * idx : hist + 2 * (H + idx);
            ar += b[k] * p[0];
            ai += b[k] * p[1];
        }
        y[2 * i]     = ar;
        y[2 * i + 1] = ai;
    }
    fir_update_hist(hist, x, ntaps, n, 2);
}

void gr4o_fir_c32_acc64(const float* b, size_t ntaps, float* hist, const float* x, double* y, size_t n) {
    const ptrdiff_t H = (ptrdiff_t)ntaps - 1;
    size_t          i, k;
    for (i = 0; i < n; ++i) {
        double ar = 0., ai = 0.;
        for (k = 0; k < ntaps; ++k) {
            const ptrdiff_t idx = (ptrdiff_t)i - (ptrdiff_t)k;
            const float*    p   = idx >= 0 ? x + 2 * idx : hist + 2 * (H + idx);
            ar += (double)b[k] * (double)p[0];
            ai += (double)b[k] * (double)p[1];
        }
        y[2 * i]     = ar;
        y[2 * i + 1] = ai;
    }
    fir_update_hist(hist, x, ntaps, n, 2);
}

/* BasicFilterProto::processBulk decimating loop (time_domain_filter.hpp:190-204) */
void gr4o_fir_decim_f32_acc64(const float* b, size_t ntaps, float* hist, const float* x, double* y, size_t n, size_t decim) {
    size_t i, k, o = 0;
    for (i = 0; i < n; ++i) {
        if (i % decim != 0) continue; /* the filter state still advances for every input: history is positional here */
        double acc = 0.;
        for (k = 0; k < ntaps; ++k) acc += (double)b[k] * (double)fir_x_f32(hist, x, ntaps, (ptrdiff_t)i - (ptrdiff_t)k);
        y[o++] = acc;
    }
    fir_update_hist(hist, x, ntaps, n, 1);
}

/* Interpolating FIR -- PARITY UNPINNED BY THE REFERENCE (it has no such block; only the rate declaration Resampling<1, L>,
 * core/include/gnuradio-4.0/annotated.hpp:121-128).  Definition of SURVEY.md Appendix A, stated literally: zero-stuff the input by L,
 * run the a1 sum (fir_filter::processOne, time_domain_filter.hpp:44-47) at the OUTPUT rate in float64, gain L.  `hist_up` is the a1
 * history at the output rate (ntaps - 1 zero-stuffed samples, oldest first), chained across calls like successive work() calls. */
void gr4o_fir_interp_f32_acc64(const float* b, size_t ntaps, size_t L, float* hist_up, const float* x, double* y, size_t n_in) {
    const size_t n_out = n_in * L;
    float*       u     = (float*)calloc(n_out ? n_out : 1, sizeof(float));
    size_t       i;
    for (i = 0; i < n_in; ++i) u[i * L] = x[i];
    gr4o_fir_f32_acc64(b, ntaps, hist_up, u, y, n_out);
    for (i = 0; i < n_out; ++i) y[i] *= (double)L;
    free(u);
}
void gr4o_fir_interp_c32_acc64(const float* b, size_t ntaps, size_t L, float* hist_up, const float* x, double* y, size_t n_in) {
    const size_t n_out = n_in * L;
    float*       u     = (float*)calloc(n_out ? 2 * n_out : 2, sizeof(float));
    size_t       i;
    for (i = 0; i < n_in; ++i) { u[2 * i * L] = x[2 * i]; u[2 * i * L + 1] = x[2 * i + 1]; }
    gr4o_fir_c32_acc64(b, ntaps, hist_up, u, y, n_out);
    for (i = 0; i < 2 * n_out; ++i) y[i] *= (double)L;
    free(u);
}

/* Decimator<T>::processBulk (time_domain_filter.hpp:234-244) */
size_t gr4o_decimate_bytes(const void* in, void* out, size_t n, size_t elem_size, size_t decim) {
    size_t i, o = 0;
    for (i = 0; i < n; ++i)
        if (i % decim == 0) memcpy((char*)out + (o++) * elem_size, (const char*)in + i * elem_size, elem_size);
    return o;
}

/* ------------------------------------------------------------------------------------------------
 * precision-templated parts (window, magnitude/phase, section step, design)
 * ---------------------------------------------------------------------------------------------- */
#define T float
#define SFX(name) name##_f32
#define SIN sinf
#define COS cosf
#define SQRT sqrtf
#define FABS fabsf
#define EXP expf
#define POW powf
#define HYPOT hypotf
#define ATAN2 atan2f
#define LOG10 log10f
#define PI_T 3.14159265358979323846f
#define EPS_T FLT_EPSILON
#define LOWEST_T (-FLT_MAX)
#define GR4O_T_IS_FLOAT 1
#include "gr4_oracle_tmpl.inc"
#undef T
#undef SFX
#undef SIN
#undef COS
#undef SQRT
#undef FABS
#undef EXP
#undef POW
#undef HYPOT
#undef ATAN2
#undef LOG10
#undef PI_T
#undef EPS_T
#undef LOWEST_T
#undef GR4O_T_IS_FLOAT

#define T double
#define SFX(name) name##_f64
#define SIN sin
#define COS cos
#define SQRT sqrt
#define FABS fabs
#define EXP exp
#define POW pow
#define HYPOT hypot
#define ATAN2 atan2
#define LOG10 log10
#define PI_T 3.14159265358979323846
#define EPS_T DBL_EPSILON
#define LOWEST_T (-DBL_MAX)
#define GR4O_T_IS_FLOAT 0
#include "gr4_oracle_tmpl.inc"
#undef T
#undef SFX

/* ------------------------------------------------------------------------------------------------
 * a3/a4: sections and cascades
 * ---------------------------------------------------------------------------------------------- */
void gr4o_section_init(gr4o_section_t* s, const double* b, int nb, const double* a, int na) {
    memset(s, 0, sizeof(*s));
    s->nb = nb;
    s->na = na;
    memcpy(s->b, b, sizeof(double) * (size_t)nb);
    memcpy(s->a, a, sizeof(double) * (size_t)na);
}

double gr4o_section_step(gr4o_section_t* s, double x, int form, int use_float) {
    return use_float ? (double)section_step_f32(s, (float)x, form) : section_step_f64(s, x, form);
}

/* Filter<T>::processOne: std::accumulate over sections (FilterTool.hpp:244-246) */
void gr4o_iir_cascade_f32(gr4o_section_t* sec, int nsec, int form, const float* x, float* y, size_t n) {
    size_t i;
    int    s;
    for (i = 0; i < n; ++i) {
        float v = x[i];
        for (s = 0; s < nsec; ++s) v = section_step_f32(&sec[s], v, form);
        y[i] = v;
    }
}

void gr4o_iir_cascade_f64(gr4o_section_t* sec, int nsec, int form, const float* x, double* y, size_t n) {
    size_t i;
    int    s;
    for (i = 0; i < n; ++i) {
        double v = (double)x[i];
        for (s = 0; s < nsec; ++s) v = section_step_f64(&sec[s], v, form);
        y[i] = v;
    }
}

/* ------------------------------------------------------------------------------------------------
 * a5: analog prototypes + frequency transforms + bilinear (FilterTool.hpp:446-624, 678-846), all float64
 * ---------------------------------------------------------------------------------------------- */
typedef struct {
    double complex poles[2 * GR4O_MAX_ORDER + 2];
    double complex zeros[2 * GR4O_MAX_ORDER + 2];
    int            np, nz;
    double         gain;
} pz_t;

void gr4o_filter_params_default(gr4o_filter_params_t* p) { /* FilterTool.hpp:66-75 */
    p->order         = 4;
    p->fLow          = NAN;
    p->fHigh         = NAN;
    p->gain          = 1.0;
    p->rippleDb      = 0.1;
    p->attenuationDb = 40;
    p->beta          = 1.6;
    p->fs            = NAN;
}

static double pz_response_mag(const pz_t* v, double omega) { /* calculateResponse<RadianPerSec,Magnitude> :459-474 */
    double complex iw = omega * I, num = 1.0, den = 1.0;
    int            i;
    for (i = 0; i < v->nz; ++i) num *= (iw - v->zeros[i]);
    for (i = 0; i < v->np; ++i) den *= (iw - v->poles[i]);
    return v->gain * cabs(num / den);
}

static int proto_butterworth(size_t order, pz_t* r) { /* :476-494 */
    size_t i;
    r->np = r->nz = 0;
    r->gain       = 1.0;
    if (order % 2 != 0) r->poles[r->np++] = -1.0;
    for (i = 0; i < order / 2; ++i) {
        double theta = M_PI * (1.0 - (double)(i * 2 + 1 + order % 2) / (2.0 * (double)order));
        double pr = cos(theta), pim = sin(theta); /* std::polar(1, theta) */
        r->poles[r->np++] = pr + pim * I;
        r->poles[r->np++] = pr - pim * I;
    }
    return 0;
}

static int proto_bessel(size_t order, pz_t* r) { /* :496-513 (tabulated, Winder 1998) */
#define C(re, im) ((re) + (im) * I)
    static const double g[11] = {1.0, 1.0, 1.6221, 2.9067, 5.1002, 11.9773, 26.8334, 41.5419, 183.3982, 306.9539, 1893.1098};
    r->nz = 0;
    switch (order) {
    case 0:
    case 1: r->np = 1; r->poles[0] = -1.0; break;
    case 2: r->np = 2; r->poles[0] = C(-1.1030, 0.6368); r->poles[1] = C(-1.1030, -0.6368); break;
    case 3: r->np = 3; r->poles[0] = -1.0509; r->poles[1] = C(-1.3270, 1.0025); r->poles[2] = C(-1.3270, -1.0025); break;
    case 4: r->np = 4; r->poles[0] = C(-1.3596, 0.4071); r->poles[1] = C(-1.3596, -0.4071); r->poles[2] = C(-0.9877, 1.2476); r->poles[3] = C(-0.9877, -1.2476); break;
    case 5: r->np = 5; r->poles[0] = -1.3851; r->poles[1] = C(-0.9606, 1.4756); r->poles[2] = C(-0.9606, -1.4756); r->poles[3] = C(-1.5069, 0.7201); r->poles[4] = C(-1.5069, -0.7201); break;
    case 6: r->np = 6; r->poles[0] = C(-1.5735, 0.3213); r->poles[1] = C(-1.5735, -0.3213); r->poles[2] = C(-1.3836, 0.9727); r->poles[3] = C(-1.3836, -0.9727); r->poles[4] = C(-0.9318, 1.6640); r->poles[5] = C(-0.9318, -1.6640); break;
    case 7: r->np = 7; r->poles[0] = -1.6130; r->poles[1] = C(-1.3797, 0.5896); r->poles[2] = C(-1.3797, -0.5896); r->poles[3] = C(-1.1397, 1.1923); r->poles[4] = C(-1.1397, -1.1923); r->poles[5] = C(-0.9104, 1.8375); r->poles[6] = C(-0.9104, -1.8375); break;
    case 8: r->np = 8; r->poles[0] = C(-1.7627, 0.2737); r->poles[1] = C(-1.7627, -0.2737); r->poles[2] = C(-0.8955, 2.0044); r->poles[3] = C(-0.8955, -2.0044); r->poles[4] = C(-1.3780, 0.8253); r->poles[5] = C(-1.3780, -0.8253); r->poles[6] = C(-1.6419, 1.3926); r->poles[7] = C(-1.6419, -1.3926); break;
    case 9: r->np = 9; r->poles[0] = -1.8081; r->poles[1] = C(-1.6532, 0.5126); r->poles[2] = C(-1.6532, -0.5126); r->poles[3] = C(-1.16532, 1.0319); r->poles[4] = C(-1.16532, -1.0319); r->poles[5] = C(-1.3683, 1.5685); r->poles[6] = C(-1.3683, -1.5685); r->poles[7] = C(-0.8788, 2.1509); r->poles[8] = C(-0.8788, -2.1509); break;
    case 10: r->np = 10; r->poles[0] = C(-1.9335, 0.2424); r->poles[1] = C(-1.9335, -0.2424); r->poles[2] = C(-0.8684, 2.2996); r->poles[3] = C(-0.8684, -2.2996); r->poles[4] = C(-1.8478, 0.7295); r->poles[5] = C(-1.8478, -0.7295); r->poles[6] = C(-1.6669, 1.2248); r->poles[7] = C(-1.6669, -1.2248); r->poles[8] = C(-1.3649, 1.7388); r->poles[9] = C(-1.3649, -1.7388); break;
    default: return -1; /* std::out_of_range */
    }
#undef C
    r->gain = g[order];
    return 0;
}

static int proto_cheby1(size_t order, double rippleDb, pz_t* r) { /* :515-531 */
    const double epsilon = sqrt(pow(10, rippleDb / 10.0) - 1);
    const double shifter = asinh(1 / epsilon) / (double)order;
    size_t       k;
    r->np = r->nz = 0;
    r->gain       = 1.0;
    for (k = 0; k < order; ++k) {
        double angle      = M_PI * ((double)k + 0.5) / (double)order;
        r->poles[r->np++] = (-sinh(shifter) * sin(angle)) + (cosh(shifter) * cos(angle)) * I;
    }
    r->gain = 1.0 / pz_response_mag(r, 0.0);
    return 0;
}

static int proto_cheby2(size_t numPoles, double stopBandDb, pz_t* r) { /* :533-562 */
    const double epsilon = 1.0 / sqrt(pow(10, stopBandDb / 10.0) - 1);
    const double v0      = asinh(1.0 / epsilon) / (double)numPoles;
    const double sinh_v0 = -sinh(v0), cosh_v0 = cosh(v0);
    size_t       k;
    r->np = r->nz = 0;
    r->gain       = 1.0;
    for (k = 1; k < numPoles; k += 2) {
        const double theta = 0.5 * ((double)k - (double)numPoles) / (double)numPoles;
        const double a = sinh_v0 * cos(M_PI * theta), b = cosh_v0 * sin(M_PI * theta), d2 = a * a + b * b;
        r->poles[r->np++] = (a / d2) + (b / d2) * I;
        r->poles[r->np++] = (a / d2) - (b / d2) * I;
        const double im   = 1.0 / cos(0.5 * M_PI * (double)k / (double)numPoles);
        r->zeros[r->nz++] = 0.0 + im * I;
        r->zeros[r->nz++] = 0.0 - im * I;
    }
    if (numPoles & 1) r->poles[r->np++] = 1.0 / sinh_v0;
    r->gain = 1.0 / pz_response_mag(r, 0.0);
    return 0;
}

static void lp_to_lp(const pz_t* proto, const gr4o_filter_params_t* p, pz_t* o) { /* :678-686 */
    int i;
    *o = *proto;
    for (i = 0; i < o->np; ++i) o->poles[i] *= 2. * M_PI * p->fLow;
    for (i = 0; i < o->nz; ++i) o->zeros[i] *= 2. * M_PI * p->fLow;
    o->gain = p->gain * proto->gain / pz_response_mag(o, 0.);
}

static void lp_to_hp(const pz_t* proto, const gr4o_filter_params_t* p, pz_t* o) { /* :688-705 */
    int i;
    *o = *proto;
    for (i = 0; i < o->np; ++i) o->poles[i] = 2. * M_PI * p->fHigh / o->poles[i];
    if (o->nz == 0) {
        for (i = 0; i < proto->np; ++i) o->zeros[i] = 0.0;
        o->nz = proto->np;
    } else {
        for (i = 0; i < o->nz; ++i) o->zeros[i] = 2. * M_PI * p->fHigh / o->zeros[i];
        while (o->nz < proto->np) o->zeros[o->nz++] = 0.0;
    }
    const double normFreq = isfinite(p->fs) ? p->fs : 10 * p->fHigh;
    o->gain               = p->gain * proto->gain / pz_response_mag(o, M_PI * normFreq);
}

static void lp_to_bp(const pz_t* proto, const gr4o_filter_params_t* p, pz_t* o) { /* :707-763 */
    const double eps = 1e-10, omega0 = 2. * M_PI * sqrt(p->fLow * p->fHigh), bw = 2. * M_PI * fabs(p->fHigh - p->fLow), Q = omega0 / bw;
    int          i;
    o->np = o->nz = 0;
    for (i = 0; i < proto->np; ++i) {
        double complex s = proto->poles[i], disc = 2.0 * omega0 * csqrt(s * s / (4.0 * Q * Q) - 1.0), base = (omega0 / Q) * s;
        o->poles[o->np++] = 0.5 * (base + disc);
        o->poles[o->np++] = 0.5 * (base - disc);
    }
    for (i = 0; i < proto->nz; ++i) {
        double complex z = proto->zeros[i];
        if (creal(z) * creal(z) + cimag(z) * cimag(z) < eps) {
            o->zeros[o->nz++] = omega0 * I;
            o->zeros[o->nz++] = -omega0 * I;
        } else {
            double complex disc = 2.0 * omega0 * csqrt(z * z / (4.0 * Q * Q) - 1.0), base = (omega0 / Q) * z;
            o->zeros[o->nz++] = 0.5 * (base + disc);
            o->zeros[o->nz++] = 0.5 * (base - disc);
        }
    }
    if (proto->np > proto->nz)
        for (i = 0; i < proto->np - proto->nz; ++i) o->zeros[o->nz++] = 0.0;
    o->gain = 1.0;
    o->gain = p->gain / pz_response_mag(o, omega0);
}

static void lp_to_bs(const pz_t* proto, const gr4o_filter_params_t* p, pz_t* o) { /* :765-819 */
    const double eps = 1e-10, omega0 = 2. * M_PI * sqrt(p->fLow * p->fHigh), bw = 2. * M_PI * fabs(p->fHigh - p->fLow), Q = omega0 / bw;
    int          i;
    o->np = o->nz = 0;
    for (i = 0; i < proto->np; ++i) {
        double complex s = proto->poles[i], disc = 0.5 * omega0 * csqrt(1.0 / (Q * Q * s * s) - 4.0), base = 0.5 * omega0 / (Q * s);
        o->poles[o->np++] = base + disc;
        o->poles[o->np++] = base - disc;
    }
    for (i = 0; i < proto->nz; ++i) {
        double complex z = proto->zeros[i];
        if (creal(z) * creal(z) + cimag(z) * cimag(z) < eps) {
            o->zeros[o->nz++] = omega0 * I;
            o->zeros[o->nz++] = -omega0 * I;
        } else {
            double complex disc = 0.5 * omega0 * csqrt(1.0 / (Q * Q * z * z) - 4.0), base = 0.5 * omega0 / (Q * z);
            o->zeros[o->nz++] = base + disc;
            o->zeros[o->nz++] = base - disc;
        }
    }
    for (i = 0; i < proto->np - proto->nz; ++i) {
        o->zeros[o->nz++] = omega0 * I;
        o->zeros[o->nz++] = -omega0 * I;
    }
    o->gain = 1.0;
    o->gain = p->gain / pz_response_mag(o, 0.);
}

/* details::sortComplexWithConjugates (:585-624) */
static int cmp_real(const void* a, const void* b) {
    double x = creal(*(const double complex*)a), y = creal(*(const double complex*)b);
    return (x > y) - (x < y);
}
static void sort_conj(double complex* v, int n) {
    const double   eps = 1e-10;
    double complex pos[2 * GR4O_MAX_ORDER + 2], neg[2 * GR4O_MAX_ORDER + 2], re[2 * GR4O_MAX_ORDER + 2];
    int            np = 0, nn = 0, nr = 0, i, o = 0;
    for (i = 0; i < n; ++i) {
        if (cimag(v[i]) > eps) pos[np++] = v[i];
        else if (cimag(v[i]) < -eps) neg[nn++] = v[i];
        else re[nr++] = v[i];
    }
    /* std::sort is not stable, but equal real parts within one sign group are conjugate-symmetric duplicates */
    qsort(pos, (size_t)np, sizeof(double complex), cmp_real);
    qsort(neg, (size_t)nn, sizeof(double complex), cmp_real);
    qsort(re, (size_t)nr, sizeof(double complex), cmp_real);
    for (i = 0; i < nn; ++i) {
        v[o++] = neg[i];
        if (i < np) v[o++] = pos[i];
    }
    for (i = nn; i < np; ++i) v[o++] = pos[i];
    for (i = 0; i < nr; ++i) v[o++] = re[i];
}

/* designAnalogFilter (:821-846) */
static int design_analog(int response, const gr4o_filter_params_t* p, int design, pz_t* analog) {
    pz_t proto;
    int  rc;
    if (p->order > GR4O_MAX_ORDER) return -1;
    switch (design) {
    case GR4O_BUTTERWORTH: rc = proto_butterworth(p->order, &proto); break;
    case GR4O_CHEBYSHEV1: rc = proto_cheby1(p->order, p->rippleDb, &proto); break;
    case GR4O_CHEBYSHEV2: rc = proto_cheby2(p->order, p->attenuationDb, &proto); break;
    case GR4O_BESSEL: rc = proto_bessel(p->order, &proto); break;
    default: return -1;
    }
    if (rc) return -1;
    if (response != GR4O_HIGHPASS && !isfinite(p->fLow)) return -1;
    if (response != GR4O_LOWPASS && !isfinite(p->fHigh)) return -1;
    switch (response) {
    case GR4O_BANDPASS: lp_to_bp(&proto, p, analog); break;
    case GR4O_BANDSTOP: lp_to_bs(&proto, p, analog); break;
    case GR4O_HIGHPASS: lp_to_hp(&proto, p, analog); break;
    default: lp_to_lp(&proto, p, analog); break;
    }
    return 0;
}

/* iir::calculateResponse<Hertz, Magnitude>(f, designAnalogFilter(...)) -- used by the reference's own design tests */
double gr4o_analog_response(int response, const gr4o_filter_params_t* p, int design, double f_hz) {
    pz_t analog;
    if (design_analog(response, p, design, &analog)) return NAN;
    return pz_response_mag(&analog, 2. * M_PI * f_hz);
}

int gr4o_iir_design(int response, const gr4o_filter_params_t* p, int design, int is_float, gr4o_section_t* sections, int cap) {
    pz_t analog;
    int  i;
    if (design_analog(response, p, design, &analog)) return -1;
    if (!isfinite(p->fs)) return -1;
    /* analogToDigitalTransform (:564-583): z = (2fs + s)/(2fs - s), no pre-warping */
    const double twoFs = 2. * p->fs;
    for (i = 0; i < analog.np; ++i) analog.poles[i] = (twoFs + analog.poles[i]) / (twoFs - analog.poles[i]);
    for (i = 0; i < analog.nz; ++i) analog.zeros[i] = (twoFs + analog.zeros[i]) / (twoFs - analog.zeros[i]);
    sort_conj(analog.poles, analog.np);
    sort_conj(analog.zeros, analog.nz);
    /* maxSectionSize: 2 for float, 4 for double (:848) */
    return is_float ? iir_sections_f32(response, p, analog.poles, analog.np, analog.zeros, analog.nz, 2, sections, cap)
                    : iir_sections_f64(response, p, analog.poles, analog.np, analog.zeros, analog.nz, 4, sections, cap);
}

int gr4o_fir_design(int response, const gr4o_filter_params_t* p, int window, int is_float, double* taps, int cap) {
    return is_float ? fir_design_f32(response, p, window, taps, cap) : fir_design_f64(response, p, window, taps, cap);
}

double gr4o_section_response(const gr4o_section_t* s, double f_norm) { return response_mag_f64(s->b, s->nb, s->a, s->na, f_norm); }

/* ------------------------------------------------------------------------------------------------
 * a8: DFT definition (algorithm/.../fourier/fft.hpp:113-153): unnormalised forward transform.
 * ---------------------------------------------------------------------------------------------- */
static int is_pow2(size_t n) { return n && !(n & (n - 1)); }

static void bitrev_permute_c64(double* d, size_t N) {
    size_t i, j = 0;
    for (i = 0; i < N; ++i) {
        if (i < j) {
            double tr = d[2 * i], ti = d[2 * i + 1];
            d[2 * i] = d[2 * j]; d[2 * i + 1] = d[2 * j + 1];
            d[2 * j] = tr; d[2 * j + 1] = ti;
        }
        size_t m = N >> 1;
        while (m >= 1 && (j & m)) { j ^= m; m >>= 1; }
        j |= m;
    }
}

void gr4o_dft_c64(const double* in, double* out, size_t N) {
    size_t k, n;
    if (N == 0) return;
    if (is_pow2(N)) {
        size_t len;
        memcpy(out, in, 2 * N * sizeof(double));
        bitrev_permute_c64(out, N);
        for (len = 2; len <= N; len <<= 1) {
            const size_t half = len >> 1;
            for (k = 0; k < N; k += len)
                for (n = 0; n < half; ++n) {
                    const double ang = -2.0 * M_PI * (double)n / (double)len; /* each twiddle from cos/sin directly */
                    const double wr = cos(ang), wi = sin(ang);
                    double* a = out + 2 * (k + n);
                    double* b = out + 2 * (k + n + half);
                    const double tr = b[0] * wr - b[1] * wi, ti = b[0] * wi + b[1] * wr;
                    b[0] = a[0] - tr; b[1] = a[1] - ti;
                    a[0] += tr; a[1] += ti;
                }
        }
        return;
    }
    for (k = 0; k < N; ++k) { /* direct O(N^2), exact angle reduction */
        long double sr = 0, si = 0;
        for (n = 0; n < N; ++n) {
            const size_t m   = (size_t)(((unsigned long long)k * n) % N);
            const double ang = -2.0 * M_PI * (double)m / (double)N;
            const double wr = cos(ang), wi = sin(ang);
            sr += (long double)(in[2 * n] * wr - in[2 * n + 1] * wi);
            si += (long double)(in[2 * n] * wi + in[2 * n + 1] * wr);
        }
        out[2 * k]     = (double)sr;
        out[2 * k + 1] = (double)si;
    }
}

/* float radix-2 DIT with a per-size twiddle table built from cos/sin of each angle (SimdFFT.hpp:419-437 accuracy class) */
static float*  g_tw32   = NULL;
static size_t  g_tw32_n = 0;
static void    ensure_tw32(size_t N) {
    size_t k;
    if (g_tw32_n == N) return;
    free(g_tw32);
    g_tw32 = (float*)malloc(N * sizeof(float)); /* N/2 complex */
    for (k = 0; k < N / 2; ++k) {
        const double ang = -2.0 * M_PI * (double)k / (double)N;
        g_tw32[2 * k]     = (float)cos(ang);
        g_tw32[2 * k + 1] = (float)sin(ang);
    }
    g_tw32_n = N;
}

void gr4o_fft_c32(const float* in, float* out, size_t N) {
    size_t i, j = 0, len, k, n;
    if (!is_pow2(N)) return;
    ensure_tw32(N);
    for (i = 0; i < N; ++i) { /* bit-reversed copy */
        out[2 * j]     = in[2 * i];
        out[2 * j + 1] = in[2 * i + 1];
        size_t m = N >> 1;
        while (m >= 1 && (j & m)) { j ^= m; m >>= 1; }
        j |= m;
    }
    for (len = 2; len <= N; len <<= 1) {
        const size_t half = len >> 1, step = N / len;
        for (k = 0; k < N; k += len)
            for (n = 0; n < half; ++n) {
                const float wr = g_tw32[2 * n * step], wi = g_tw32[2 * n * step + 1];
                float* a = out + 2 * (k + n);
                float* b = out + 2 * (k + n + half);
                const float tr = b[0] * wr - b[1] * wi, ti = b[0] * wi + b[1] * wr;
                b[0] = a[0] - tr; b[1] = a[1] - ti;
                a[0] += tr; a[1] += ti;
            }
    }
}

/* ------------------------------------------------------------------------------------------------
 * a7: FFT block processBulk (blocks/fourier/.../fft.hpp:147-171): window -> FFT -> mag(shift) / phase(shift) / Re / Im
 * ---------------------------------------------------------------------------------------------- */
void gr4o_fft_block_c32(const float* frame, size_t N, int window, int in_db, int in_deg, int unwrap, float* mag, float* phase, float* re, float* im) {
    float* w  = (float*)malloc(N * sizeof(float));
    float* in = (float*)malloc(2 * N * sizeof(float));
    float* sp = (float*)malloc(2 * N * sizeof(float));
    size_t i;
    gr4o_window_f32(window, w, N, 1.6f); /* fft.hpp:141 create(_window, _windowType) -> default beta */
    for (i = 0; i < N; ++i) { /* fft.hpp:155-162 */
        in[2 * i]     = frame[2 * i] * w[i];
        in[2 * i + 1] = frame[2 * i + 1] * w[i];
    }
    gr4o_fft_c32(in, sp, N);
    gr4o_magnitude_f32(sp, N, mag, 0, in_db, 1);
    gr4o_phase_f32(sp, N, phase, 0, in_deg, unwrap, 1);
    for (i = 0; i < N; ++i) { /* fft.hpp:217-220: Re/Im are NOT shifted */
        re[i] = sp[2 * i];
        im[i] = sp[2 * i + 1];
    }
    free(w); free(in); free(sp);
}

void gr4o_fft_block_c32_truth(const float* frame, size_t N, int window, int in_db, int in_deg, int unwrap, double* mag, double* phase, double* re, double* im) {
    float*  w  = (float*)malloc(N * sizeof(float));
    double* in = (double*)malloc(2 * N * sizeof(double));
    double* sp = (double*)malloc(2 * N * sizeof(double));
    size_t  i;
    gr4o_window_f32(window, w, N, 1.6f); /* the block's window values are floats (value_type) */
    for (i = 0; i < N; ++i) {
        in[2 * i]     = (double)frame[2 * i] * (double)w[i];
        in[2 * i + 1] = (double)frame[2 * i + 1] * (double)w[i];
    }
    gr4o_dft_c64(in, sp, N);
    gr4o_magnitude_f64(sp, N, mag, 0, in_db, 1);
    gr4o_phase_f64(sp, N, phase, 0, in_deg, unwrap, 1);
    for (i = 0; i < N; ++i) { re[i] = sp[2 * i]; im[i] = sp[2 * i + 1]; }
    free(w); free(in); free(sp);
}

void gr4o_fft_block_f32_truth(const float* frame, size_t N, int window, int in_db, int in_deg, int unwrap, double* mag, double* phase, double* re, double* im) {
    /* real input: full Hermitian spectrum of N bins (algorithm fft.hpp:245-255), block keeps N/2 mag/phase values
     * (fft.hpp:142-143, computeHalfSpectrum) and Re/Im of the LAST N/2 entries of the (1+N/2)-sized _outData (fft.hpp:221-227).
     * NOTE: compute() resizes _outData to in.size() == N, so .last(N/2) are bins N/2..N-1. */
    float*  w  = (float*)malloc(N * sizeof(float));
    double* in = (double*)malloc(2 * N * sizeof(double));
    double* sp = (double*)malloc(2 * N * sizeof(double));
    size_t  i, h = N / 2;
    gr4o_window_f32(window, w, N, 1.6f);
    for (i = 0; i < N; ++i) { in[2 * i] = (double)frame[i] * (double)w[i]; in[2 * i + 1] = 0.0; }
    gr4o_dft_c64(in, sp, N);
    gr4o_magnitude_f64(sp, N, mag, 1, in_db, 1);
    gr4o_phase_f64(sp, N, phase, 1, in_deg, unwrap, 1);
    for (i = 0; i < h; ++i) { re[i] = sp[2 * (h + i)]; im[i] = sp[2 * (h + i) + 1]; }
    free(w); free(in); free(sp);
}

/* ------------------------------------------------------------------------------------------------
 * headline chain (BASELINE.json configs[1]): fir_filter on complex<float> -> FFT block frames -> |X|^2
 * ---------------------------------------------------------------------------------------------- */
void gr4o_chain_c32(const float* b, size_t ntaps, float* hist, size_t N, int window, const float* x, float* mag2, size_t n) {
    const size_t frames = n / N;
    float*       y  = (float*)malloc(2 * N * sizeof(float));
    float*       sp = (float*)malloc(2 * N * sizeof(float));
    float*       w  = (float*)malloc(N * sizeof(float));
    size_t       f, i;
    gr4o_window_f32(window, w, N, 1.6f);
    for (f = 0; f < frames; ++f) {
        gr4o_fir_c32(b, ntaps, hist, x + 2 * f * N, y, N);
        for (i = 0; i < N; ++i) { y[2 * i] *= w[i]; y[2 * i + 1] *= w[i]; }
        gr4o_fft_c32(y, sp, N);
        for (i = 0; i < N; ++i) mag2[f * N + i] = sp[2 * i] * sp[2 * i] + sp[2 * i + 1] * sp[2 * i + 1];
    }
    free(y); free(sp); free(w);
}

void gr4o_chain_c32_truth(const float* b, size_t ntaps, float* hist, size_t N, int window, const float* x, double* mag2, size_t n) {
    const size_t frames = n / N;
    double*      y  = (double*)malloc(2 * N * sizeof(double));
    double*      sp = (double*)malloc(2 * N * sizeof(double));
    float*       w  = (float*)malloc(N * sizeof(float));
    size_t       f, i;
    gr4o_window_f32(window, w, N, 1.6f);
    for (f = 0; f < frames; ++f) {
        gr4o_fir_c32_acc64(b, ntaps, hist, x + 2 * f * N, y, N);
        for (i = 0; i < N; ++i) { y[2 * i] *= (double)w[i]; y[2 * i + 1] *= (double)w[i]; }
        gr4o_dft_c64(y, sp, N);
        for (i = 0; i < N; ++i) mag2[f * N + i] = sp[2 * i] * sp[2 * i] + sp[2 * i + 1] * sp[2 * i + 1];
    }
    free(y); free(sp); free(w);
}

/* ------------------------------------------------------------------------------------------------
 * a11/a12: math blocks.  op()(a, value) on T operands: integer promotion, then narrowing back to T (Math.hpp:54).
 * Signed overflow is computed through the unsigned type (two's complement wrap) to stay defined in C.
 * ---------------------------------------------------------------------------------------------- */
size_t gr4o_dtype_size(int dtype) {
    static const size_t s[14] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 8, 16, 8, 16};
    return (dtype >= 0 && dtype < 14) ? s[dtype] : 0;
}

#define INT_OP(TYPE, UTYPE)                                                      \
    static TYPE op_##TYPE(int op, TYPE a, TYPE b) {                              \
        switch (op) {                                                            \
        case GR4O_ADD: return (TYPE)((UTYPE)a + (UTYPE)b);                       \
        case GR4O_SUB: return (TYPE)((UTYPE)a - (UTYPE)b);                       \
        case GR4O_MUL: return (TYPE)((UTYPE)a * (UTYPE)b);                       \
        default: return (TYPE)(b == 0 ? 0 : a / b); /* div-by-zero is UB in the reference; defined as 0 here */ \
        }                                                                        \
    }
INT_OP(uint8_t, uint32_t)
INT_OP(uint16_t, uint32_t)
INT_OP(uint32_t, uint32_t)
INT_OP(uint64_t, uint64_t)
INT_OP(int8_t, uint32_t)
INT_OP(int16_t, uint32_t)
INT_OP(int32_t, uint32_t)
INT_OP(int64_t, uint64_t)
/* note: for 8/16-bit types the reference promotes to int; (U32)a op (U32)b truncated to 8/16 bits gives the identical
 * low bits for + - *; division of promoted ints equals division in the narrow type (a / b above). */

static float  op_float(int op, float a, float b) { return op == GR4O_ADD ? a + b : op == GR4O_SUB ? a - b : op == GR4O_MUL ? a * b : a / b; }
static double op_double(int op, double a, double b) { return op == GR4O_ADD ? a + b : op == GR4O_SUB ? a - b : op == GR4O_MUL ? a * b : a / b; }
static float complex op_c32(int op, float complex a, float complex b) { return op == GR4O_ADD ? a + b : op == GR4O_SUB ? a - b : op == GR4O_MUL ? a * b : a / b; }
static double complex op_c64(int op, double complex a, double complex b) { return op == GR4O_ADD ? a + b : op == GR4O_SUB ? a - b : op == GR4O_MUL ? a * b : a / b; }

/* gr::UncertainValue<float | double> (meta/include/gnuradio-4.0/meta/UncertainValue.hpp:34-40): {value, uncertainty}.  MathOpImpl<UncertainValue<T>, op> applies
 * op()(a, value) with BOTH operands uncertain and a real value type: the "both ValueType[T,U] are arithmetic uncertainties" branches of operator+ (:121-133),
 * operator- (:159-171), operator* (:192-204) and operator/ (:221-243): uncorrelated propagation, std::hypot for the combination, every product and quotient in T. */
typedef struct { float v, u; } gr4o_uf32;
typedef struct { double v, u; } gr4o_uf64;
static gr4o_uf32 op_uf32(int op, gr4o_uf32 a, gr4o_uf32 b) {
    gr4o_uf32 r;
    switch (op) {
    case GR4O_ADD: r.v = a.v + b.v; r.u = hypotf(a.u, b.u); break;
    case GR4O_SUB: r.v = a.v - b.v; r.u = hypotf(a.u, b.u); break;
    case GR4O_MUL: r.v = a.v * b.v; r.u = hypotf(a.v * b.u, b.v * a.u); break;
    default: r.v = a.v / b.v; r.u = hypotf(a.u / b.v, b.u * a.v / (b.v * b.v)); break;
    }
    return r;
}
static gr4o_uf64 op_uf64(int op, gr4o_uf64 a, gr4o_uf64 b) {
    gr4o_uf64 r;
    switch (op) {
    case GR4O_ADD: r.v = a.v + b.v; r.u = hypot(a.u, b.u); break;
    case GR4O_SUB: r.v = a.v - b.v; r.u = hypot(a.u, b.u); break;
    case GR4O_MUL: r.v = a.v * b.v; r.u = hypot(a.v * b.u, b.v * a.u); break;
    default: r.v = a.v / b.v; r.u = hypot(a.u / b.v, b.u * a.v / (b.v * b.v)); break;
    }
    return r;
}

#define APPLY(TYPE, FN)                                                                           \
    do {                                                                                          \
        const TYPE* a_ = (const TYPE*)a; const TYPE* b_ = (const TYPE*)b; TYPE* o_ = (TYPE*)out;    \
        for (i = 0; i < n; ++i) o_[i] = FN(op, a_[i], b_[b_stride ? i : 0]);                       \
    } while (0)

static int math_binary(int op, int dtype, const void* a, const void* b, int b_stride, void* out, size_t n) {
    size_t i;
    switch (dtype) {
    case GR4O_U8: APPLY(uint8_t, op_uint8_t); break;
    case GR4O_U16: APPLY(uint16_t, op_uint16_t); break;
    case GR4O_U32: APPLY(uint32_t, op_uint32_t); break;
    case GR4O_U64: APPLY(uint64_t, op_uint64_t); break;
    case GR4O_I8: APPLY(int8_t, op_int8_t); break;
    case GR4O_I16: APPLY(int16_t, op_int16_t); break;
    case GR4O_I32: APPLY(int32_t, op_int32_t); break;
    case GR4O_I64: APPLY(int64_t, op_int64_t); break;
    case GR4O_F32: APPLY(float, op_float); break;
    case GR4O_F64: APPLY(double, op_double); break;
    case GR4O_C32: APPLY(float complex, op_c32); break;
    case GR4O_C64: APPLY(double complex, op_c64); break;
    case GR4O_UF32: APPLY(gr4o_uf32, op_uf32); break;
    case GR4O_UF64: APPLY(gr4o_uf64, op_uf64); break;
    default: return -1;
    }
    return 0;
}

int gr4o_math_const(int op, int dtype, const void* in, void* out, size_t n, const void* value) { /* Math.hpp:38-56 */
    return math_binary(op, dtype, in, value, 0, out, n);
}

int gr4o_math_nary(int op, int dtype, const void* const* ins, size_t n_inputs, void* out, size_t n) { /* Math.hpp:100-107 */
    size_t k;
    if (n_inputs == 0) return -1;
    memmove(out, ins[0], n * gr4o_dtype_size(dtype)); /* std::copy(ins[0]) */
    for (k = 1; k < n_inputs; ++k)
        if (math_binary(op, dtype, out, ins[k], 1, out, n)) return -1; /* std::transform(sout, ins[n], sout, op{}) */
    return 0;
}

/* ------------------------------------------------------------------------------------------------
 * a13: Rotator::processOne (blocks/math/.../Rotator.hpp:51-61): increment FIRST, single +-2pi wrap, then multiply.
 * ---------------------------------------------------------------------------------------------- */
void gr4o_rotator_c32(float* phase_state, float inc, const float* x, float* y, size_t n) {
    const float two_pi = 2.0f * 3.14159265358979323846f;
    float       ph     = *phase_state;
    size_t      i;
    for (i = 0; i < n; ++i) {
        ph += inc;
        if (ph > two_pi) ph -= two_pi;
        else if (ph < 0.0f) ph += two_pi;
        const float c = cosf(ph), s = sinf(ph);
        const float xr = x[2 * i], xi = x[2 * i + 1];
        y[2 * i]     = xr * c - xi * s; /* std::complex operator* (finite operands) */
        y[2 * i + 1] = xr * s + xi * c;
    }
    *phase_state = ph;
}

void gr4o_rotator_c64(double* phase_state, double inc, const double* x, double* y, size_t n) {
    const double two_pi = 2.0 * M_PI;
    double       ph     = *phase_state;
    size_t       i;
    for (i = 0; i < n; ++i) {
        ph += inc;
        if (ph > two_pi) ph -= two_pi;
        else if (ph < 0.0) ph += two_pi;
        const double c = cos(ph), s = sin(ph);
        const double xr = x[2 * i], xi = x[2 * i + 1];
        y[2 * i]     = xr * c - xi * s;
        y[2 * i + 1] = xr * s + xi * c;
    }
    *phase_state = ph;
}
