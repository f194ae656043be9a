/*
 * gr4_oracle.h -- CPU ORACLE for the MI355X hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This is a plain-C restatement of the arithmetic of fair-acc/gnuradio4's
 * blocks/filter, blocks/fourier, blocks/math hot path.  Only
 * tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it;
 * the product (libgr4hip.so, gnuradio4_amd/, include/) never links, imports or calls it.
 *
 * Pinning status (see DESIGN.md "Oracle"):
 *   - Xoshiro256pp / GaussianNoise: checked bit-for-bit against the REFERENCE ITSELF
 *     (oracle/_ref/libgr4ref.so, compiled from the reference's own headers where they lie)
 *     and against the seed-0 known-answer vector of algorithm/test/qa_Xoshiro256pp.cpp:55-69.
 *   - window / FFT / magnitude / unwrap / FIR / IIR / math / rotator: checked against the
 *     golden vectors and known-answer tests the reference's own qa_*.cpp hold
 *     (tests/golden/reference_vectors.json, transcribed data only).
 *   - std::transform_reduce(std::execution::unseq) summation ORDER is unspecified in the
 *     reference, so bit-level parity of float FIR/IIR sums is unpinned; parity for float
 *     paths is defined against the float64 variants below (SURVEY.md section 7 "Parity definition").
 *
 * All reference citations are relative to /root/reference.
 * Complex values are interleaved {re, im} pairs.
 */
#ifndef GR4_ORACLE_H
#define GR4_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- a15: rng (algorithm/.../rng/Xoshiro256pp.hpp:22-96, GaussianNoise.hpp:59-99) ---- */
typedef struct { uint64_t s[4]; } gr4o_xoshiro_t;
void     gr4o_xoshiro_seed(gr4o_xoshiro_t* r, uint64_t seed);
uint64_t gr4o_xoshiro_next(gr4o_xoshiro_t* r);
void     gr4o_gauss_fill_f32(gr4o_xoshiro_t* r, float* out, size_t n, float amplitude, float offset);
void     gr4o_gauss_fill_c32(gr4o_xoshiro_t* r, float* out_interleaved, size_t n, float amplitude, float offset);
/* test-signal recipe of SURVEY.md 8(d): unit-power complex noise (seed) + tone at f_rel, amplitude a */
void gr4o_signal_c32(uint64_t seed, float* out_interleaved, size_t n, double tone_frel, double tone_amp, float noise_amp);
void gr4o_signal_f32(uint64_t seed, float* out, size_t n, double tone_frel, double tone_amp, float noise_amp);

/* ---- a1/a2: fir_filter<T>::processOne (time_domain_filter.hpp:44-47) ----
 * y[n] = sum_k b[k] x[n-k]; zero initial history; `hist` holds the previous ntaps-1 inputs
 * (oldest first) and is updated so calls can be chained like successive work() calls.
 * _f32: float accumulation in natural k order (what a scalar build of the reference does);
 * _f64: float64 accumulation of the same float inputs (the parity truth). */
void gr4o_fir_f32(const float* b, size_t ntaps, float* hist, const float* x, float* y, size_t n);
void gr4o_fir_f32_acc64(const float* b, size_t ntaps, float* hist, const float* x, double* y, size_t n);
void gr4o_fir_c32(const float* b, size_t ntaps, float* hist_interleaved, const float* x, float* y, size_t n);
void gr4o_fir_c32_acc64(const float* b, size_t ntaps, float* hist_interleaved, const float* x, double* y, size_t n);
/* decimating variant: BasicFilterProto::processBulk (time_domain_filter.hpp:190-204): filter every
 * input, keep outputs with i % decim == 0 (i restarts per call; n must be a multiple of decim). */
void gr4o_fir_decim_f32_acc64(const float* b, size_t ntaps, float* hist, const float* x, double* y, size_t n, size_t decim);
/* interpolating FIR: PARITY UNPINNED BY THE REFERENCE (no such block upstream; SURVEY.md Appendix A definition): zero-stuff by L, the a1 sum at
 * the output rate in float64, gain L.  hist_up: ntaps - 1 samples at the OUTPUT rate (zero-stuffed), chained across calls. */
void gr4o_fir_interp_f32_acc64(const float* b, size_t ntaps, size_t L, float* hist_up, const float* x, double* y, size_t n_in);
void gr4o_fir_interp_c32_acc64(const float* b, size_t ntaps, size_t L, float* hist_up_interleaved, const float* x, double* y, size_t n_in);
/* Decimator<T>::processBulk (time_domain_filter.hpp:234-244): keep i % decim == 0; bytes-exact copy */
size_t gr4o_decimate_bytes(const void* in, void* out, size_t n, size_t elem_size, size_t decim);

/* ---- a3/a4: detail::computeFilter 4 forms (FilterTool.hpp:116-158), Filter<T>::processOne cascade
 *      (FilterTool.hpp:244-246), iir_filter<T,form>::processOne (time_domain_filter.hpp:89-121) ---- */
enum { GR4O_DF_I = 0, GR4O_DF_II = 1, GR4O_DF_I_T = 2, GR4O_DF_II_T = 3 };
#define GR4O_MAX_ORDER 16
typedef struct {
    int    nb, na;                  /* coefficient counts, a[0] assumed 1 (time_domain_filter.hpp:95) */
    double b[GR4O_MAX_ORDER + 1], a[GR4O_MAX_ORDER + 1];
    double xh[GR4O_MAX_ORDER + 1];  /* inputHistory, newest first  */
    double yh[GR4O_MAX_ORDER + 1];  /* outputHistory, newest first */
} gr4o_section_t;
void   gr4o_section_init(gr4o_section_t* s, const double* b, int nb, const double* a, int na);
/* one sample through one section; `use_float` rounds every product/sum to float like T=float */
double gr4o_section_step(gr4o_section_t* s, double x, int form, int use_float);
/* cascade over nsec sections, n samples (std::accumulate over sections, FilterTool.hpp:244-246) */
void gr4o_iir_cascade_f32(gr4o_section_t* sec, int nsec, int form, const float* x, float* y, size_t n);
void gr4o_iir_cascade_f64(gr4o_section_t* sec, int nsec, int form, const float* x, double* y, size_t n);

/* ---- a5: filter design (FilterTool.hpp:415-423, 476-917, 964-1071) ---- */
enum { GR4O_LOWPASS = 0, GR4O_HIGHPASS = 1, GR4O_BANDPASS = 2, GR4O_BANDSTOP = 3 };
enum { GR4O_BUTTERWORTH = 0, GR4O_BESSEL = 1, GR4O_CHEBYSHEV1 = 2, GR4O_CHEBYSHEV2 = 3 };
typedef struct {
    size_t order; double fLow, fHigh, gain, rippleDb, attenuationDb, beta, fs;
} gr4o_filter_params_t;
void gr4o_filter_params_default(gr4o_filter_params_t* p);
/* fir::designFilter<T> ; is_float selects T=float arithmetic. returns tap count (<= cap) or -1 */
int gr4o_fir_design(int response, const gr4o_filter_params_t* p, int window, int is_float, double* taps, int cap);
/* iir::designFilter<T> -> biquad (float) / 4th-order (double) sections. returns section count or -1 */
int gr4o_iir_design(int response, const gr4o_filter_params_t* p, int design, int is_float, gr4o_section_t* sections, int cap);
/* analog prototype response in Hz (FilterTool.hpp:459-474 on designAnalogFilter :821-846) */
double gr4o_analog_response(int response, const gr4o_filter_params_t* p, int design, double f_hz);
/* calculateResponse<Normalised, Magnitude> for one section (FilterTool.hpp:379-413), double */
double gr4o_section_response(const gr4o_section_t* s, double f_norm);

/* ---- a10: window::create (window.hpp:69-183); type ids follow window.hpp:35 ---- */
enum { GR4O_WIN_NONE = 0, GR4O_WIN_RECT, GR4O_WIN_HAMMING, GR4O_WIN_HANN, GR4O_WIN_HANNEXP, GR4O_WIN_BLACKMAN,
       GR4O_WIN_NUTTALL, GR4O_WIN_BLACKMANHARRIS, GR4O_WIN_BLACKMANNUTTALL, GR4O_WIN_FLATTOP, GR4O_WIN_EXPONENTIAL,
       GR4O_WIN_KAISER };
int gr4o_window_f32(int type, float* w, size_t n, float beta);
int gr4o_window_f64(int type, double* w, size_t n, double beta);

/* ---- a8: unnormalised forward DFT X[k] = sum x[n] e^{-2 pi i k n / N} (algorithm/.../fft.hpp:113-153) ----
 * _f64: float64 truth (radix-2 for powers of two, Bluestein-free direct O(N^2) otherwise);
 * _f32: float radix-2 with per-twiddle cos/sin (SimdFFT-like accuracy, SimdFFT.hpp:419-437). */
void gr4o_dft_c64(const double* in_interleaved, double* out_interleaved, size_t N);
void gr4o_fft_c32(const float* in_interleaved, float* out_interleaved, size_t N); /* N power of two */

/* ---- a9: fft_common.hpp:20-56 / 71-89 / 91-123 ---- */
void gr4o_magnitude_f32(const float* spec, size_t N, float* mag, int half, int in_db, int shift);
void gr4o_magnitude_f64(const double* spec, size_t N, double* mag, int half, int in_db, int shift);
void gr4o_unwrap_f64(double* phase, size_t n);
void gr4o_unwrap_f32(float* phase, size_t n);
void gr4o_phase_f32(const float* spec, size_t N, float* ph, int half, int in_deg, int unwrap, int shift);
void gr4o_phase_f64(const double* spec, size_t N, double* ph, int half, int in_deg, int unwrap, int shift);

/* ---- a7: FFT block processBulk (blocks/fourier/.../fft.hpp:147-171) for complex<float> input:
 * window -> FFT -> magnitude(shift) + phase(shift) + Re + Im. Outputs each N floats. 64-bit truth variant too. */
void gr4o_fft_block_c32(const float* frame, size_t N, int window, int in_db, int in_deg, int unwrap,
                        float* mag, float* phase, float* re, float* im);
void gr4o_fft_block_c32_truth(const float* frame, size_t N, int window, int in_db, int in_deg, int unwrap,
                              double* mag, double* phase, double* re, double* im);
/* real float input: outputs N/2 values (fft.hpp:140-142, 218-224) */
void gr4o_fft_block_f32_truth(const float* frame, size_t N, int window, int in_db, int in_deg, int unwrap,
                              double* mag, double* phase, double* re, double* im);

/* ---- headline chain (BASELINE.json configs[1]): cf32 FIR -> N-pt FFT (window) -> mag2, natural bin order.
 * mag2[k] = Re^2+Im^2 ; relation to the reference block output pinned in tests.
 * _f32: reference-faithful float path (CPU baseline); _truth: float64 everywhere. n multiple of N. */
void gr4o_chain_c32(const float* b, size_t ntaps, float* hist, size_t N, int window,
                    const float* x, float* mag2, size_t n);
void gr4o_chain_c32_truth(const float* b, size_t ntaps, float* hist, size_t N, int window,
                          const float* x, double* mag2, size_t n);

/* ---- a11/a12: math blocks (blocks/math/.../Math.hpp:38-56, 100-107) ----
 * dtype ids: 0 u8,1 u16,2 u32,3 u64,4 i8,5 i16,6 i32,7 i64,8 f32,9 f64,10 c32,11 c64. op: 0 add,1 sub,2 mul,3 div */
enum { GR4O_U8 = 0, GR4O_U16, GR4O_U32, GR4O_U64, GR4O_I8, GR4O_I16, GR4O_I32, GR4O_I64, GR4O_F32, GR4O_F64, GR4O_C32, GR4O_C64, GR4O_UF32, GR4O_UF64 /* UncertainValue<float|double>: {value, uncertainty} pairs */ };
enum { GR4O_ADD = 0, GR4O_SUB, GR4O_MUL, GR4O_DIV };
size_t gr4o_dtype_size(int dtype);
int    gr4o_math_const(int op, int dtype, const void* in, void* out, size_t n, const void* value);
int    gr4o_math_nary(int op, int dtype, const void* const* ins, size_t n_inputs, void* out, size_t n);

/* ---- a13: Rotator<complex<float>>::processOne (Rotator.hpp:51-61); phase state in/out ---- */
void gr4o_rotator_c32(float* phase_state, float phase_inc, const float* x, float* y, size_t n);
void gr4o_rotator_c64(double* phase_state, double phase_inc, const double* x, double* y, size_t n);

#ifdef __cplusplus
}
#endif
#endif
