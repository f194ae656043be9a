// fir.hip -- time-domain FIR / polyphase decimating FIR / Decimator kernels for gfx950.
//
// Replaces gr::filter::fir_filter<T>::processOne (blocks/filter/.../time_domain_filter.hpp:44-47; per-sample
// HistoryBuffer::push_front + K-term transform_reduce driven by Block.hpp:1723-1761) and the decimating
// BasicFilterProto::processBulk (:190-204) by one launch per work() span:
//   * the input span (+ the K-1 history samples carried in HBM between launches) is staged once into LDS,
//     de-interleaved by polyphase branch when decim > 1, so every input sample is read from HBM exactly once;
//   * each lane owns 8 consecutive output floats and slides a 12-float register window over its LDS row:
//     one ds_read_b128 of samples + one broadcast ds_read of taps feeds 32 (real) / 16 (complex) FMAs;
//   * complex<float> data x real taps is the same kernel on the interleaved float view with a tap stride of 2.
// Bound: FP32 FMA rate (2K flop per real output sample), not HBM -- see DESIGN.md "fir_poly".
// Long spans leave this kernel: float 33..256 taps and polyphase decimators go to the MFMA kernels of fir_batched.hip, complex <= 256 taps
// to the frequency-domain kernel of chain_fused.hip (gr4hip_fir_process below decides).
#include "common.hpp"
#include "fir_exact.hpp"
#include "ewise.hpp"
#include "fir_window.hpp"
#include "fir_f16_common.hpp" // (hf_wave_sum)

namespace gr4 {

// x: n_in samples; hist: the last `hcap` input samples before x (oldest first); tp: [D][Qpad] phase-major taps
// (tp[p][q] = b[q*D + p], zero padded); y: n_out samples, y[m] = sum_k b[k] x[m*D - k].
// HOOK: the filter's per-sample neighbours ride in this launch (ewise.hpp).  `pre` is applied to every sample of x on its way into LDS (pre.pos = absolute
// stream index of x[0]); the carried history holds what the prologue made of earlier samples and is taken as it lies (the next history this launch writes is
// made of prologue OUTPUTS too) -- exactly the filter's memory when the blocks run one after the other; `post` is applied to every output sample before its
// store (post.pos = absolute index of y[0]).
// gthr > 0: the workgroup judges its own outputs like the matrix-pipe kernels do (fir_f16.hip): output power (of its quietest wave) below gthr x the power of the samples it
// was given = the filter removes nearly everything, where ANY float32 sum -- this one phase by phase, the reference's tap by tap -- shows its rounding against the output
// (measured, complex decimate-by-2, tone near fs / 2: 12 x the reference order's error).  Such a workgroup evaluates its outputs again from the samples it has staged, with
// float64 products and sums (one rounding at the end): no second launch, hooks included (the staged samples are the prologue's outputs).  Non-finite power either side
// compares false: those outputs keep the float32 sums and with them the reference's classes.
template <int S, int BS, bool HOOK>
__global__ __launch_bounds__(BS) void fir_poly_kernel(const float* __restrict__ x, const float* __restrict__ hist, const float* __restrict__ tp,
                                                       float* __restrict__ y, long n_in, long n_out, int hcap, int D, int G, float* __restrict__ new_hist,
                                                       EwiseHook pre, EwiseHook post, float gthr) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    __shared__ float judge_red[2 * (BS / 64)];
    constexpr int R  = kFirR;
    constexpr int E  = 4 / S;
    const int     Lf = BS * R + 4 * G; // floats per phase row
    const int     Qp = G * E;          // padded taps per phase
    float*        xl = smem;           // [D][Lf]
    float*        bl = smem + (size_t)D * Lf; // [D][Qp]
    const int     tid   = threadIdx.x;
    const long    M0    = (long)blockIdx.x * (BS * R / S); // first output sample of this block
    const long    jbase = M0 - (4 * G) / S;                // first phase-stream sample held in LDS
    const int     Ls    = Lf / S;                          // samples per phase row

    for (int i = tid; i < D * Qp; i += BS) bl[i] = tp[i];

    // ---- stage inputs: u enumerates input samples i = i_lo + u; sample i belongs to phase p = (-i) mod D at j = ceil(i/D)
    const long i_lo  = jbase * D - (D - 1);
    const long total = (long)Ls * D;
    if (D == 1 && S == 1) {
        for (long u4 = (long)tid * 4; u4 < total; u4 += (long)BS * 4) { // i_lo is a multiple of 4 here
            const long i = i_lo + u4;
            float4     v;
            if (i >= 0 && i + 3 < n_in) {
                v = *reinterpret_cast<const float4*>(x + i);
                if constexpr (HOOK) {
                    float e[4] = {v.x, v.y, v.z, v.w};
                    ewise_hook<float, 4>(e, pre, i);
                    v = make_float4(e[0], e[1], e[2], e[3]);
                }
            } else {
                float t[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const long ii = i + c;
                    t[c] = (ii >= 0) ? (ii < n_in ? x[ii] : 0.f) : (ii >= -(long)hcap ? hist[hcap + ii] : 0.f);
                    if constexpr (HOOK)
                        if (ii >= 0 && ii < n_in) t[c] = ewise_hook1<float>(t[c], pre, ii);
                }
                v = make_float4(t[0], t[1], t[2], t[3]);
            }
            *reinterpret_cast<float4*>(xl + u4) = v;
        }
    } else if (D == 1 && S == 2) {
        for (long u2 = (long)tid * 2; u2 < total; u2 += (long)BS * 2) { // two complex samples = one float4
            const long i = i_lo + u2;
            float4     v;
            if (i >= 0 && i + 1 < n_in) {
                v = *reinterpret_cast<const float4*>(x + 2 * i);
                if constexpr (HOOK) {
                    float2 e[2] = {make_float2(v.x, v.y), make_float2(v.z, v.w)};
                    ewise_hook<float2, 2>(e, pre, i);
                    v = make_float4(e[0].x, e[0].y, e[1].x, e[1].y);
                }
            } else {
                float t[4];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const long   ii = i + c;
                    const float* p  = (ii >= 0) ? (ii < n_in ? x + 2 * ii : nullptr) : (ii >= -(long)hcap ? hist + 2 * (hcap + ii) : nullptr);
                    float2       q  = p ? make_float2(p[0], p[1]) : make_float2(0.f, 0.f);
                    if constexpr (HOOK)
                        if (p && ii >= 0) q = ewise_hook1<float2>(q, pre, ii);
                    t[2 * c]     = q.x;
                    t[2 * c + 1] = q.y;
                }
                v = make_float4(t[0], t[1], t[2], t[3]);
            }
            *reinterpret_cast<float4*>(xl + 2 * u2) = v;
        }
    } else {
        for (long u = tid; u < total; u += BS) {
            const long   i  = i_lo + u;
            const int    jl = (int)(u / D);
            const int    p  = D - 1 - (int)(u % D);
            const float* src = (i >= 0) ? (i < n_in ? x + S * i : nullptr) : (i >= -(long)hcap ? hist + S * (hcap + i) : nullptr);
            float*       dst = xl + (size_t)p * Lf + (size_t)jl * S;
            if constexpr (HOOK) {
                if (src && i >= 0) {
                    if constexpr (S == 1) dst[0] = ewise_hook1<float>(src[0], pre, i);
                    else { const float2 q = ewise_hook1<float2>(make_float2(src[0], src[1]), pre, i); dst[0] = q.x; dst[S - 1] = q.y; }
                    continue;
                }
            }
#pragma unroll
            for (int c = 0; c < S; ++c) dst[c] = src ? src[c] : 0.f;
        }
    }
    __syncthreads();

    float acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) acc[r] = 0.f;
    const int c0 = tid * R + 4 * G; // local float index of this lane's first output sample's x[m*D] position
    for (int p = 0; p < D; ++p) {
        const float* row = xl + (size_t)p * Lf;
        const float* tg  = bl + (size_t)p * Qp;
        float        w[12];
#pragma unroll
        for (int q = 0; q < 3; ++q) {
            const float4 v = *reinterpret_cast<const float4*>(row + c0 - 4 + 4 * q);
            w[4 * q] = v.x; w[4 * q + 1] = v.y; w[4 * q + 2] = v.z; w[4 * q + 3] = v.w;
        }
        int g = 0;
        for (; g + 3 <= G; g += 3) { // window rotation sequence ROT = 0,2,1 (three steps return to ROT 0)
            fir_step<S, 0>(acc, w, tg + (g + 0) * E, row + c0 - 4 * (g + 0) - 8, (g + 1) < G);
            fir_step<S, 2>(acc, w, tg + (g + 1) * E, row + c0 - 4 * (g + 1) - 8, (g + 2) < G);
            fir_step<S, 1>(acc, w, tg + (g + 2) * E, row + c0 - 4 * (g + 2) - 8, (g + 3) < G);
        }
        if (g < G) {
            fir_step<S, 0>(acc, w, tg + g * E, row + c0 - 4 * g - 8, (g + 1) < G);
            if (g + 1 < G) fir_step<S, 2>(acc, w, tg + (g + 1) * E, row + c0 - 4 * (g + 1) - 8, false);
        }
    }

    if (gthr > 0.f) {
        float px = 0.f, py = 0.f;
        for (int p = 0; p < D; ++p) { // the samples this lane's outputs sit on: every staged sample of the workgroup's own span exactly once
            const float4 a = *reinterpret_cast<const float4*>(xl + (size_t)p * Lf + c0), b = *reinterpret_cast<const float4*>(xl + (size_t)p * Lf + c0 + 4);
            px += (a.x * a.x + a.y * a.y) + (a.z * a.z + a.w * a.w) + (b.x * b.x + b.y * b.y) + (b.z * b.z + b.w * b.w);
        }
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (M0 * S + (long)tid * R + r < n_out * S) py = fmaf(acc[r], acc[r], py); // (outputs past the span do not exist: the filter's transient into the zero-staged samples behind a truncated tone is not output power)
        px = hf_wave_sum(px);
        py = hf_wave_sum(py);
        if ((tid & 63) == 0) { judge_red[tid >> 6] = px; judge_red[BS / 64 + (tid >> 6)] = py; }
        __syncthreads();
        //  only waves whose 64 R outputs all exist vote: a wave past n_out sees zero-staged samples, its output power is 0 and the quietest-wave
        // statistic sent the last workgroup of EVERY call -- and every call shorter than 3/4 of a workgroup -- through the float64 loop.  A workgroup without one
        // complete wave (a span shorter than 64 R outputs) is judged on wave 0 alone, whose input sum counts the real samples only.
        px = 0.f;
        py = 3.4e38f;
        int  nc = 0;
        bool finite = true;
#pragma unroll
        for (int w = 0; w < BS / 64; ++w)
            if (M0 * S + (long)(w + 1) * 64 * R <= n_out * S) {
                ++nc;
                px += judge_red[w];
                py = fminf(py, judge_red[BS / 64 + w]); // (fminf drops a NaN: the test below sees it through px, or the wave's outputs are NaN anyway)
                finite = finite && judge_red[BS / 64 + w] < 3.0e38f;
            }
        if (nc == 0) { nc = 1; px = judge_red[0]; py = judge_red[BS / 64]; finite = py < 3.0e38f; }
        if (finite && py * (float)nc * (float)D < gthr * px) { // (uniform over the workgroup)
            double a64[R];
#pragma unroll
            for (int r = 0; r < R; ++r) a64[r] = 0.0;
            for (int p = 0; p < D; ++p) {
                const float* row = xl + (size_t)p * Lf + c0;
                const float* tg  = bl + (size_t)p * Qp;
                for (int q = 0; q < Qp; ++q) { // tap b[q D + p] meets the phase-p sample q places back
                    const double t = (double)tg[q];
#pragma unroll
                    for (int r = 0; r < R; ++r) a64[r] = __builtin_fma(t, (double)row[r - q * S], a64[r]);
                }
            }
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r] = (float)a64[r];
        }
    }
    const long of    = M0 * S + (long)tid * R; // first output float of this lane
    const long nf    = n_out * S;
    if constexpr (HOOK) {
        if (post.n_ops > 0) {
            if constexpr (S == 1) ewise_hook<float, R>(acc, post, of);
            else {
                float2 e[R / 2];
#pragma unroll
                for (int r = 0; r < R / 2; ++r) e[r] = make_float2(acc[2 * r], acc[2 * r + 1]);
                ewise_hook<float2, R / 2>(e, post, of / 2);
#pragma unroll
                for (int r = 0; r < R / 2; ++r) { acc[2 * r] = e[r].x; acc[2 * r + 1] = e[r].y; }
            }
        }
    }
    if (of + R <= nf) {
        *reinterpret_cast<float4*>(y + of)     = make_float4(acc[0], acc[1], acc[2], acc[3]);
        *reinterpret_cast<float4*>(y + of + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
    } else {
#pragma unroll
        for (int r = 0; r < R; ++r)
            if (of + r < nf) y[of + r] = acc[r];
    }
    // a span served by this launch alone: workgroup 0 also writes the block's next history (the other half of the ping-pong pair; nobody reads it in this
    // launch), which saves the separate update launch -- a scheduler's 4 Ki .. 64 Ki-sample work() chunks cost launches, not arithmetic
    if (new_hist != nullptr && blockIdx.x == 0) {
        if constexpr (HOOK) { // the next history is made of what the prologue produces
            for (int h = tid; h < hcap; h += BS) {
                const long i = n_in - hcap + h;
                if constexpr (S == 1) new_hist[h] = i >= 0 ? ewise_hook1<float>(x[i], pre, i) : hist[hcap + i];
                else {
                    const float2 q = i >= 0 ? ewise_hook1<float2>(make_float2(x[2 * i], x[2 * i + 1]), pre, i) : make_float2(hist[2 * (hcap + i)], hist[2 * (hcap + i) + 1]);
                    new_hist[2 * h]     = q.x;
                    new_hist[2 * h + 1] = q.y;
                }
            }
        } else {
            for (int t = tid; t < hcap * S; t += BS) {
                const long h = t / S, c = t % S, i = n_in - hcap + h;
                new_hist[t]  = (i >= 0) ? x[i * S + c] : hist[(hcap + i) * S + c];
            }
        }
    }
}

// the shape no tiled kernel takes (decimation x taps beyond every LDS tiling, e.g. complex data decimated by 100): one output per lane, the reference's sum term for term
// (time_domain_filter.hpp:44-47 -- taps ascending, float32 fma), samples through the caches.  Slow and always there.
template <int S>
__global__ __launch_bounds__(256) void fir_generic_kernel(const float* __restrict__ x, const float* __restrict__ hist, int hcap, const float* __restrict__ b, int ntaps, long D,
                                                          float* __restrict__ y, long n_out, long n_in) {
    const long m = (long)blockIdx.x * 256 + threadIdx.x;
    if (m >= n_out) return;
    float acc[S];
#pragma unroll
    for (int c = 0; c < S; ++c) acc[c] = 0.f;
    for (int k = 0; k < ntaps; ++k) {
        const long   i = m * D - k;
        const float* p = i >= 0 ? (i < n_in ? x + i * S : nullptr) : (i >= -(long)hcap ? hist + ((long)hcap + i) * S : nullptr);
        if (p == nullptr) continue;
#pragma unroll
        for (int c = 0; c < S; ++c) acc[c] = fmaf(b[k], p[c], acc[c]);
    }
#pragma unroll
    for (int c = 0; c < S; ++c) y[m * S + c] = acc[c];
}

// new_hist[h] = virtual_input[n_in - hcap + h]  (virtual_input(i<0) = old_hist[hcap + i])
__global__ void fir_hist_update_kernel(const float* __restrict__ x, const float* __restrict__ old_hist, float* __restrict__ new_hist, long n_in,
                                       int hcap, int S) {
    const long t = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= (long)hcap * S) return;
    const long h = t / S, c = t % S;
    const long i = n_in - hcap + h;
    new_hist[t]  = (i >= 0) ? x[i * S + c] : old_hist[(hcap + i) * S + c];
}

// the stored history changes its meaning (a prologue gain moves into or out of the taps): every sample times `factor`
__global__ void fir_hist_scale_kernel(float* __restrict__ hist, int nfloats, float factor) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < nfloats) hist[t] *= factor;
}

template <typename V>
__global__ void decimate_kernel(const V* __restrict__ in, V* __restrict__ out, long n_out, long decim) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < n_out; m += stride) out[m] = in[m * decim];
}

} // namespace gr4

namespace gr4 { // chain_fused.hip: fast-convolution path for long complex inputs
struct ChainFused;
int  chain_fused_create(ChainFused** out, const float* taps, size_t ntaps, size_t fft_size, int window, int algo);
int  chain_fused_fir(ChainFused* c, const float* d_in, const float* d_hist256, size_t n_frames, float* d_y, hipStream_t st);
void chain_fused_set_measure(ChainFused* c, bool on);
int  chain_fused_power_ratio(ChainFused* c, bool wait, bool fir_output, float* ratio, float* marked_fraction = nullptr, float* float64_fraction = nullptr);
void chain_fused_destroy(ChainFused* c);

// fir_decim_fd.hip: decimate-by-8 real FIR (<= 1024 taps) as overlap-save blocks in the frequency domain
struct FirDecimFd;
int  fir_decim_fd_supported(size_t ntaps, size_t decim);
int  fir_decim_fd_create(FirDecimFd** out, const float* taps, size_t ntaps);
void fir_decim_fd_destroy(FirDecimFd* c);
struct DecimFdIir { const float* tab; int nsec; const float* state_in; float* state_out; int warm_blocks; };
int  fir_decim_fd_run(FirDecimFd* c, const float* d_in, size_t n_in, const float* d_hist, int hcap, float* d_out, hipStream_t st, bool measure, const DecimFdIir* iir = nullptr);
int  fir_decim_fd_power_ratio(FirDecimFd* c, bool wait, float* ratio);

// fir_batched.hip: block-Toeplitz FIR on the f32 MFMA units (real, <= 256 taps)
void fir_mfma_make_afrag(const float* taps, size_t ntaps, size_t nch, int* Kp_out, int* KS_out, std::vector<float>* af_out);
int  fir_mfma_launch(int KS, const float* x, long in_stride, const float* hist, const float* afrag, float* y, long out_stride, long n, unsigned nch, hipStream_t st, float* new_hist);
int  fir_mfma_c32_launch(int KS, const float* x, long n, const float* hist, const float* afrag, float* y, hipStream_t st, float* new_hist);
void fir_decim_band_make_row(const float* taps, size_t ntaps, size_t D, int* Kp_out, std::vector<float>* row);
void fir_bf16_make_afrag(const float* taps, size_t ntaps, int* KS_out, std::vector<unsigned short>* af, size_t nch, int force_ks);
int  fir_bf16_launch(int KS, const float* x, long n, const float* hist, int Kh, const void* afrag, float* y, hipStream_t st, float* new_hist, long in_stride, long out_stride, unsigned nch, int delay, int accum);
bool fir_f16_make_afrag(const float* taps, size_t ntaps, int* KS_out, std::vector<unsigned short>* af, size_t nch, int force_ks); // fir_f16.hip
bool fir_decim_f16_make_table(const float* taps, size_t ntaps, size_t D, int* KQ_out, std::vector<unsigned short>* tab, bool cplx); // fir_decim_f16.hip
int  fir_decim_f16_launch(int D, int KQ, const float* x, long n_in, const float* hist, int Kh, const void* table, float* y, long n_out, hipStream_t st, float* new_hist, int guard, unsigned char* flags, int cplx,
                          const EwiseHook* pre, const EwiseHook* post);
int  fir_f16_c32_launch(int KS, const float* x, long n, const float* hist, int Kh, const void* table, float* y, hipStream_t st, float* new_hist, int guard, unsigned char* flags, int delay = 0, int accum = 0, float gthr = 0.f);
int  fir_f16_launch(int KS, const float* x, long n, const float* hist, int Kh, const void* table, float* y, hipStream_t st, float* new_hist, long in_stride, long out_stride, unsigned nch, int delay, int accum, int guard,
                    unsigned char* flags, long flags_stride, float gthr);
int  fir_bf16_c32_launch(int KS, const float* x, long n, const float* hist, int Kh, const void* afrag, float* y, hipStream_t st, float* new_hist);
void fir_decim_bf16_make_afrag(const float* taps, size_t ntaps, size_t D, int* KS_out, int* Hb_out, std::vector<unsigned short>* af, bool cplx);
int  fir_decim_bf16_launch(int KS, int D, int Hb, const float* x, long n_in, const float* hist, int Kh, const void* afrag, float* y, long n_out, hipStream_t st, float* new_hist,
                           const EwiseHook* pre, const EwiseHook* post, bool cplx, unsigned char* flags, float gthr, int* seg_out);
int  fir_decim_band_launch(int D, int Kp, const float* x, const float* hist, int hcap, const float* row, float* y, long n_out, long n_in, hipStream_t st);
void fir_mfma_make_afrag_decim(const float* taps, size_t ntaps, size_t D, int* Kp_out, int* KS_out, std::vector<float>* af_out);
int  fir_mfma_decim_launch(int KS, int D, const float* x, const float* hist, const float* afrag, float* y, long n_out, hipStream_t st);

// out[h] = sample at stream position -len + h (h < len), from the hcap-sample history (zeros before it)
template <typename T>
__global__ void fir_hist_widen_kernel(const T* __restrict__ hist, int hcap, T* __restrict__ out, int len) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h < len) out[h] = h >= len - hcap ? hist[h - (len - hcap)] : T{};
}
} // namespace gr4

using namespace gr4;

constexpr size_t kFdFrame     = 8192;
// the per-segment guard of the split-product kernels (fir_f16.hip, fir_decim_f16.hip, fir_bf16.hip's decimators): a segment is marked for the second evaluation when its output
// power is below this x (sum b^2) x its input power -- 21 dB more rejected than white noise would lose.  Where it comes from: the two-term f16 products' error is ~1.3e-7 rms of
// the PRODUCTS' level sqrt(sum b^2) rms(x), its largest values ~4 x that; at an output 2^-3.5 = 1 / 11.3 of that level they reach 6e-6 of the output -- inside the 1e-5 bar with
// margin (measured with tools/tone_ratio.py: at the 36 dB threshold of round 4 unmarked segments reached 4.4e-5).  The tables of fir_f16.hip / fir_decim_f16.hip carry the same number.
constexpr double kGuardSegmentRatio = 1.0 / 128.0;
constexpr size_t kMfmaMinSamples = 1 << 16;
constexpr size_t kFdMinFrames = 64; // below this the direct-form kernel is as fast (the persistent FD grid wants >= 1 frame per CU)

struct gr4hip_fir {
    int                dtype = GR4HIP_F32;
    int                S     = 1;
    size_t             ntaps = 0, decim = 1;
    size_t             hcap  = 32; // HistoryBuffer capacity (time_domain_filter.hpp:36: inputHistory{32})
    std::vector<float> taps;
    int                G = 0;      // tap groups per phase
    DeviceBuffer       d_taps;     // [D][Qpad]
    DeviceBuffer       d_tapsf;    // the taps as they are (what fir_exact_kernel multiplies with)
    double             tap_power = 0; // sum b^2 of `taps` (the guard's thresholds are multiples of it)
    double             guard_ratio = kGuardSegmentRatio; // (library-internal callers may tighten it: the chain's kernel pair squares this filter's output -- chain.hip)
    DeviceBuffer       d_flags_fd; // the same for the frequency-domain kernels' spans (judged behind their launch at their own thresholds)
    DeviceBuffer       d_flags;    // one byte per segment of the f16 matrix-pipe kernels' last launch: the segments fir_exact_kernel evaluates again behind it
    DeviceBuffer       d_hist[2];  // ping-pong history (hcap samples each)
    int                cur = 0;
    int                algo = GR4HIP_FIR_AUTO; // gr4hip_fir_set_algo
    bool               f32_user = false;       // (what gr4hip_fir_set_algo asked for: a reset re-arms the guard and goes back to it)
    bool               bf16_user = false;      // GR4HIP_FIR_TIME_DOMAIN_BF16X3: the three-term bf16 products where the default would take the two-term f16 ones
    bool               f32_products = false;   // GR4HIP_FIR_TIME_DOMAIN_F32, or a stream the dynamic-range guard has moved: no three-term bf16 products (the f32 matrix-pipe kernels instead)
    int                guard_mode = GR4HIP_GUARD_STRICT; // gr4hip_fir_set_guard_mode
    gr4::ChainFused*   fd  = nullptr; // complex, decim 1, ntaps <= 256: frequency-domain plan (created on first use)
    // dynamic-range guard of GR4HIP_FIR_AUTO (same policy as the chain's, include/gr4hip.h): the fast convolution's error floor is ~2e-6 of the INPUT rms;
    // below an output / input power ratio of 0.04 the direct form takes over (first use: probed synchronously; later: from finished measurements)
    bool               fd_probed = false, fd_blocked = false;
    float              fd_ratio  = -1.f;
    gr4::FirDecimFd*   dfd = nullptr; // float, decim 8, <= 1024 taps: frequency-domain decimator (created on first use)
    DeviceBuffer       d_band;        // float, decim >= 10: tap row of fir_decim_band_kernel (built on first use)
    int                bandKp = 0;
    DeviceBuffer       d_bfrag;       // float / complex, 65 .. 256 taps (float: slices of 256 up to 2048): the three bf16 tap-fragment tables of fir_mfma_bf16x3_kernel (built on first use)
    int                bfKS = 0;
    DeviceBuffer       d_hfrag;       // float, 33 .. 256 taps and the 256-tap slices of 384 .. 1024: the two-term f16 tables of fir_mfma_f16x2_kernel (fir_f16.hip; built on first use)
    int                hfKS = 0;      // < 0: taps that form cannot carry
    DeviceBuffer       d_dhtab;       // float, decimate by 8, 169 .. 1025 taps: the table of fir_decim8_f16x2_kernel (fir_decim_f16.hip; built on first use)
    int                dhKQ = 0;      // < 0: a shape that kernel does not carry
    std::vector<size_t> hf_off;       // per slice: offset into d_hfrag (in 16-bit units) ...
    std::vector<int>    hf_ks;        // ... and window size
    DeviceBuffer       d_bdfrag;      // float, decim 2 .. 9, short branches: band-form bf16 fragments (fir_decim_bf16x3_kernel)
    int                bdKS = 0, bdHb = 0; // bdKS < 0: the window does not fit that kernel
    std::vector<size_t> bf_off;       // per slice: offset into d_bfrag (in bf16 elements) ...
    std::vector<int>    bf_ks;        // ... and window size
    DeviceBuffer       d_hist256, d_histc;
    DeviceBuffer       d_afrag;       // real, decim 1, 32 < ntaps <= 256: MFMA A fragments (built on first use)
    int                mKp = 0, mKS = 0;
    // per-sample neighbours executed in this filter's launch (gr4hip_fir_set_prologue / _epilogue): private copies of the programs
    gr4hip_ewise*      pre = nullptr, *post = nullptr;
    bool               pre_is_gain = false, post_is_gain = false; // nothing but real gains: folded into the taps (a FIR filter is linear)
    double             pre_gain = 1.0, post_gain = 1.0;
    double             folded = 1.0;   // the gain the device tap tables currently carry: `taps` = folded x `user_taps`
    std::vector<float> user_taps;      // the block's setting `b` as given
    long               pos = 0;        // input samples consumed since create / reset: the absolute index a rotator op's phase is a function of
    double             hist_gain = 1.0; // the stored history times this = what the prologue made of those samples (a folded prologue gain keeps RAW samples in the
                                        // history -- the taps carry the gain --, a hooked prologue keeps its outputs there: 1)
    DeviceBuffer       d_mid;          // gr4hip_fir_iir_process, two-launch path: the decimated stream between the filter and the cascade
    DeviceBuffer       d_pre;          // long spans of filters that take a matrix-pipe kernel: the prologue's output (one element-wise launch in front of it)
    // the stream rule (common.hpp): create / reset / set_taps / set_prologue only note what the device state has to become; fir_state_on() enqueues it on the stream of
    // the next call that needs it, behind whatever that stream still has in flight for this handle
    std::vector<float> tp_host;        // d_taps' image ([D][Qpad]) as fir_upload_taps laid it out
    bool               taps_dirty = false; // d_taps / d_tapsf do not hold tp_host / taps yet
    bool               zero_hist  = true;  // d_hist[cur] is to be zeroed (create, reset, a history that had to grow)
    double             hist_rescale = 1.0; // d_hist[cur] is to be multiplied by this (a prologue gain that moved into or out of the taps)
    ~gr4hip_fir() { delete pre; delete post; }
};

static size_t bit_ceil_sz(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }

static int fir_upload_taps(gr4hip_fir* f) {
    const size_t D = f->decim, K = f->ntaps;
    const int    E = 4 / f->S;
    const size_t Q = ceil_div(K, D);
    f->G           = (int)ceil_div(Q, (size_t)E);
    const size_t Qp = (size_t)f->G * E;
    std::vector<float>& tp = f->tp_host;
    tp.assign(D * Qp, 0.f);
    for (size_t k = 0; k < K; ++k) tp[(k % D) * Qp + (k / D)] = f->taps[k];
    int rc = f->d_taps.ensure(tp.size() * sizeof(float)); // (growing frees the old table: hipFree waits for the device, nothing in flight loses its taps)
    if (rc) return rc;
    rc = f->d_tapsf.ensure(K * sizeof(float));
    if (rc) return rc;
    f->tap_power = 0;
    for (float b : f->taps) f->tap_power += (double)b * b;
    f->taps_dirty = true; // uploaded by fir_state_on, on the stream of the next call
    return GR4HIP_OK;
}

static int fir_alloc_hist(gr4hip_fir* f) {
    const size_t bytes = f->hcap * f->S * sizeof(float);
    for (int k = 0; k < 2; ++k) {
        int rc = f->d_hist[k].ensure(bytes);
        if (rc) return rc;
    }
    f->zero_hist    = true; // zeroed by fir_state_on, on the stream of the next call (a launch still in flight may be writing either half of the pair)
    f->hist_rescale = 1.0;
    return GR4HIP_OK;
}
// The handle's device state as the call about to be enqueued on `st` must see it: pending tap upload, pending zeroing / rescaling of the carried history.
// Everything goes onto `st`, so it is ordered behind the earlier launches of this handle on that stream (Block.hpp:606, 916-917: reset() / settingsChanged() run
// between two work() calls of the block's own worker) and in front of the launch that follows.
static int fir_state_on(gr4hip_fir* f, hipStream_t st) {
    if (f->taps_dirty) {
        GR4_HIP_TRY(hipMemcpyAsync(f->d_taps.ptr, f->tp_host.data(), f->tp_host.size() * sizeof(float), hipMemcpyHostToDevice, st));
        GR4_HIP_TRY(hipMemcpyAsync(f->d_tapsf.ptr, f->taps.data(), f->ntaps * sizeof(float), hipMemcpyHostToDevice, st));
        f->taps_dirty = false;
    }
    if (f->zero_hist) {
        GR4_HIP_TRY(hipMemsetAsync(f->d_hist[f->cur].ptr, 0, f->hcap * f->S * sizeof(float), st));
        f->zero_hist    = false;
        f->hist_rescale = 1.0;
    }
    if (f->hist_rescale != 1.0) {
        const int nf = (int)(f->hcap * f->S);
        hipLaunchKernelGGL(fir_hist_scale_kernel, dim3((unsigned)ceil_div(nf, 256)), dim3(256), 0, st, (float*)f->d_hist[f->cur].ptr, nf, (float)f->hist_rescale);
        GR4_LAUNCH_CHECK();
        f->hist_rescale = 1.0;
    }
    return GR4HIP_OK;
}

struct FirHooks { EwiseHook pre, post; bool any = false; };

template <int S, int BS, bool HOOK>
static int fir_launch_h(const gr4hip_fir* f, const float* x, const float* hist, float* y, long n_in, long n_out, size_t lds, hipStream_t st, float* new_hist, const FirHooks& hk) {
    auto kern = fir_poly_kernel<S, BS, HOOK>;
    if (lds > 48 * 1024) GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long TOs  = BS * kFirR / S;
    const long grid = ceil_div(n_out, TOs);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(BS), lds, st, x, hist, (const float*)f->d_taps.ptr, y, n_in, n_out,
                       (int)f->hcap, (int)f->decim, f->G, new_hist, hk.pre, hk.post, f->guard_mode != GR4HIP_GUARD_OFF && f->ntaps > 1 ? (float)(f->tap_power * f->guard_ratio) : 0.f);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}
template <int S, int BS>
static int fir_launch(const gr4hip_fir* f, const float* x, const float* hist, float* y, long n_in, long n_out, size_t lds, hipStream_t st, float* new_hist, const FirHooks& hk) {
    return hk.any ? fir_launch_h<S, BS, true>(f, x, hist, y, n_in, n_out, lds, st, new_hist, hk) : fir_launch_h<S, BS, false>(f, x, hist, y, n_in, n_out, lds, st, new_hist, hk);
}

// every device table derived from the taps is rebuilt on next use
static void fir_invalidate_tables(gr4hip_fir* f) {
    if (f->fd) { chain_fused_destroy(f->fd); f->fd = nullptr; }
    if (f->dfd) { fir_decim_fd_destroy(f->dfd); f->dfd = nullptr; }
    f->mKS = 0;
    f->bandKp = 0;
    f->bfKS = 0;
    f->hfKS = 0;
    f->dhKQ = 0;
    f->bdKS = 0;
}
// `taps` = gain x `user_taps` (float64 product, rounded once), uploaded
static int fir_apply_gain(gr4hip_fir* f, double gain) {
    f->taps.resize(f->user_taps.size());
    for (size_t k = 0; k < f->taps.size(); ++k) f->taps[k] = (float)(gain * (double)f->user_taps[k]);
    f->folded = gain;
    fir_invalidate_tables(f);
    return fir_upload_taps(f);
}
// the stored history follows a change of its meaning: hist_gain -> g (a prologue gain that moves into or out of the taps)
static int fir_rescale_history(gr4hip_fir* f, double g) {
    if (g == f->hist_gain) return GR4HIP_OK;
    if (f->pos > 0) f->hist_rescale *= f->hist_gain / g; // applied by fir_state_on on the next call's stream (in float64 until then: two changes in a row round once)
    f->hist_gain = g;
    return GR4HIP_OK;
}
// what this call does with the neighbours: gains ride in the taps, the rest are hooks (or, in front of / behind a matrix-pipe kernel, element-wise launches)
static int fir_prepare_hooks(gr4hip_fir* f, FirHooks* hk, hipStream_t st) {
    const bool   fold_pre = f->pre && f->pre_is_gain, fold_post = f->post && f->post_is_gain;
    const double want     = (fold_pre ? f->pre_gain : 1.0) * (fold_post ? f->post_gain : 1.0);
    if (want != f->folded) { if (const int rc = fir_apply_gain(f, want)) return rc; }
    if (f->pre && !fold_pre) {
        f->pre->pos = f->pos;
        if (const int rc = ewise_device_ops(f->pre, &hk->pre, st)) return rc;
    }
    if (f->post && !fold_post) {
        f->post->pos = f->pos / (long)f->decim;
        if (const int rc = ewise_device_ops(f->post, &hk->post, st)) return rc;
    }
    hk->any = hk->pre.n_ops > 0 || hk->post.n_ops > 0;
    return GR4HIP_OK;
}
static int fir_set_hook(gr4hip_fir* f, const gr4hip_ewise_t* prog, bool prologue) {
    GR4_REQUIRE(f, "fir_set_%s: null handle", prologue ? "prologue" : "epilogue");
    if (prog && prog->dtype != f->dtype) { set_error("fir_set_%s: the program's dtype %d is not the filter's (%d)", prologue ? "prologue" : "epilogue", prog->dtype, f->dtype); return GR4HIP_UNSUPPORTED; }
    gr4hip_ewise* copy = nullptr;
    if (prog && !prog->user.empty()) {
        copy = ewise_clone(prog);
        GR4_REQUIRE(copy, "out of host memory");
    }
    if (prologue) {
        // the history keeps what the OUTGOING prologue made of the samples in it (the filter's memory when the blocks run one after the other): with a hooked
        // prologue it holds those outputs already, with a folded gain it holds raw samples and is brought to the new prologue's convention here
        double     g       = 1.0;
        const bool is_gain = copy && ewise_as_real_gain(copy, &g);
        if (const int rc = fir_rescale_history(f, is_gain ? g : 1.0)) { delete copy; return rc; }
        delete f->pre;
        f->pre         = copy;
        f->pre_is_gain = is_gain;
        f->pre_gain    = is_gain ? g : 1.0;
    } else {
        delete f->post;
        f->post         = copy;
        f->post_is_gain = copy && ewise_as_real_gain(copy, &f->post_gain);
    }
    return GR4HIP_OK;
}

extern "C" {

int gr4hip_fir_create(gr4hip_fir_t** out, int dtype, const float* h_taps, size_t ntaps, size_t decim) {
    GR4_REQUIRE(out, "fir: null output handle");
    GR4_REQUIRE(dtype == GR4HIP_F32 || dtype == GR4HIP_C32, "fir: dtype must be F32 or C32 (got %d)", dtype);
    GR4_REQUIRE(h_taps && ntaps >= 1, "fir: need at least one tap");
    GR4_REQUIRE(decim >= 1, "fir: decim must be >= 1");
    auto* f = new (std::nothrow) gr4hip_fir();
    GR4_REQUIRE(f, "out of host memory");
    f->dtype = dtype;
    f->S     = dtype == GR4HIP_C32 ? 2 : 1;
    f->decim = decim;
    f->ntaps = ntaps;
    f->taps.assign(h_taps, h_taps + ntaps);
    f->user_taps = f->taps;
    if (ntaps > f->hcap) f->hcap = bit_ceil_sz(ntaps); // time_domain_filter.hpp:38-42
    int rc = fir_upload_taps(f);
    if (!rc) rc = fir_alloc_hist(f);
    if (rc) { delete f; return rc; }
    *out = f;
    return GR4HIP_OK;
}

int gr4hip_fir_set_taps(gr4hip_fir_t* f, const float* h_taps, size_t ntaps) {
    GR4_REQUIRE(f && h_taps && ntaps >= 1, "fir_set_taps: bad arguments");
    f->user_taps.assign(h_taps, h_taps + ntaps);
    f->ntaps = ntaps;
    f->fd_probed = f->fd_blocked = false;
    f->f32_products = f->f32_user;
    f->fd_ratio  = -1.f;
    int rc   = fir_apply_gain(f, f->folded); // every table derived from the taps is rebuilt on next use
    if (rc) return rc;
    if (ntaps > f->hcap) { // the reference replaces the HistoryBuffer (history is lost) only when it must grow
        f->hcap = bit_ceil_sz(ntaps);
        return fir_alloc_hist(f);
    }
    return GR4HIP_OK;
}

int gr4hip_fir_reset(gr4hip_fir_t* f) {
    GR4_REQUIRE(f, "fir_reset: null handle");
    f->fd_probed = f->fd_blocked = false;
    f->f32_products = f->f32_user;
    f->fd_ratio  = -1.f;
    f->pos = 0;
    return fir_alloc_hist(f);
}

int gr4hip_fir_set_prologue(gr4hip_fir_t* f, const gr4hip_ewise_t* prog) { return fir_set_hook(f, prog, true); }
int gr4hip_fir_set_epilogue(gr4hip_fir_t* f, const gr4hip_ewise_t* prog) { return fir_set_hook(f, prog, false); }

int gr4hip_fir_set_algo(gr4hip_fir_t* f, int algo) {
    GR4_REQUIRE(f, "fir_set_algo: null handle");
    GR4_REQUIRE(algo >= GR4HIP_FIR_AUTO && algo <= GR4HIP_FIR_TIME_DOMAIN_BF16X3, "fir_set_algo: unknown algo %d", algo);
    f->algo         = algo == GR4HIP_FIR_TIME_DOMAIN_F32 ? (int)GR4HIP_FIR_TIME_DOMAIN : (algo == GR4HIP_FIR_TIME_DOMAIN_BF16X3 ? (f->S == 2 ? (int)GR4HIP_FIR_TIME_DOMAIN : (int)GR4HIP_FIR_AUTO) : algo); // (float filters take their matrix-pipe kernels under AUTO)
    f->f32_products = f->f32_user = algo == GR4HIP_FIR_TIME_DOMAIN_F32;
    f->bf16_user    = algo == GR4HIP_FIR_TIME_DOMAIN_BF16X3;
    return GR4HIP_OK;
}
int gr4hip_fir_set_guard_mode(gr4hip_fir_t* f, int mode) {
    GR4_REQUIRE(f, "fir_set_guard_mode: null handle");
    GR4_REQUIRE(mode >= GR4HIP_GUARD_STRICT && mode <= GR4HIP_GUARD_OFF, "fir_set_guard_mode: unknown mode %d", mode);
    f->guard_mode = mode;
    if (f->fd) chain_fused_set_measure(f->fd, mode != GR4HIP_GUARD_OFF && f->ntaps > 1);
    return GR4HIP_OK;
}
// the three-term bf16 kernels are off for this handle (GR4HIP_FIR_EXACT_F32: IEEE float32 multiply-add, the reference's Inf / NaN behaviour) or for the process (developer switch)
static bool no_bf16x3(const gr4hip_fir_t* f) { return f->algo == GR4HIP_FIR_EXACT_F32 || f->f32_products || dev_switch(kDevFirNoBf16x3); }

// does this call take the band-form bf16 decimators (fir_bf16.hip)?  Builds their fragments on first use.
// does the f16 band-form decimator (fir_decim_f16.hip: D = 4 / 8 / 16 / 32, two-term f16 products, carries the filter's programs since round 5) take this call?
static bool fir_decim_f16_shape(const gr4hip_fir_t* f, size_t n_in, const void* d_in, const void* d_out, bool hooked) {
    if (hooked && f->decim == 4 && (f->S == 2 ? 2 * f->ntaps - 1 : f->ntaps) > 577) return false; // (decimate by 4 with more than five K-steps per wave AND programs: 256 registers do not hold it without spills; the bf16 band kernel takes it)
    // developer knobs.  Measured (G input samples/s, bf16 band kernel / this one), float decimate by 8: 64 taps 1037 / 998, 100 taps 984 / 999, 128 taps 963 / 998
    static const size_t kMin8   = [] { const char* e = std::getenv("GR4HIP_FIR_DECIM_F16_MIN_TAPS"); return e ? (size_t)std::atoi(e) : (size_t)97; }();
    static const size_t kMinW   = [] { const char* e = std::getenv("GR4HIP_FIR_DECIM_F16_MIN_TAPS_WIDE"); return e ? (size_t)std::atoi(e) : (size_t)33; }(); // (decimate by 16 / 32)
    static const size_t kMinC8  = [] { const char* e = std::getenv("GR4HIP_FIR_DECIM_F16_MIN_TAPS_C8"); return e ? (size_t)std::atoi(e) : (size_t)64; }();   // (complex, decimate by 8)
    static const size_t kMin4   = [] { const char* e = std::getenv("GR4HIP_FIR_DECIM_F16_MIN_TAPS_D4"); return e ? (size_t)std::atoi(e) : (size_t)33; }();   // (decimate by 4, float and complex: round 5)
    const size_t D = f->decim;
    const bool   taps_ok = (f->S == 1 && ((D == 8 && f->ntaps >= kMin8) || ((D == 16 || D == 32) && f->ntaps >= kMinW) || (D == 4 && f->ntaps >= kMin4))) ||
                         (f->S == 2 && (((D == 16 || D == 32) && f->ntaps >= kMinW) || (D == 8 && f->ntaps >= kMinC8) || (D == 4 && f->ntaps >= kMin4)));
    return taps_ok && f->ntaps <= 1025 && n_in * f->S >= (1u << 17) && ((reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(d_in)) & 15) == 0 && f->algo == GR4HIP_FIR_AUTO && !f->f32_user &&
           !f->bf16_user && !dev_switch(kDevFirNoBf16x3) && !dev_switch(kDevFirNoF16x2) && !dev_switch(kDevFirNoDecimF16) && f->dhKQ >= 0;
}
static bool fir_decim_bf16_ready(gr4hip_fir_t* f, size_t n_in, const void* d_in, const void* d_out, int* rc, hipStream_t st) {
    *rc = GR4HIP_OK;
    if (!(f->decim >= (f->S == 1 ? (size_t)2 : (size_t)3) && f->decim <= (f->S == 1 ? (size_t)12 : (size_t)16) && n_in / f->decim >= (1u << 14) && f->algo == GR4HIP_FIR_AUTO && f->bdKS >= 0 &&
          ((reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(d_in)) & 15) == 0 && !no_bf16x3(f)))
        return false;
    if (f->bdKS == 0) {
        std::vector<unsigned short> af;
        int                         ks = 0;
        fir_decim_bf16_make_afrag(f->taps.data(), f->ntaps, f->decim, &ks, &f->bdHb, &af, f->S == 2);
        if (ks == 0 || (f->S == 1 && f->decim == 8 && ks > 9 && f->ntaps <= 1024)) f->bdKS = -1; // the window does not fit -- or decimate-by-8 with a long window, where the
                                                                                                  // frequency-domain kernel is faster (762 against 305 G input samples/s at 1024 taps)
        else {
            *rc = f->d_bdfrag.ensure(af.size() * sizeof(unsigned short));
            if (!*rc) { hipError_t e = hipMemcpyAsync(f->d_bdfrag.ptr, af.data(), af.size() * sizeof(unsigned short), hipMemcpyHostToDevice, st); if (e != hipSuccess) { set_error("fir: upload failed: %s", hipGetErrorString(e)); *rc = GR4HIP_RUNTIME_ERROR; } }
            if (*rc) return false;
            f->bdKS = ks;
        }
    }
    return f->bdKS > 0;
}
static int fir_process_core(gr4hip_fir_t* f, const void* d_in, size_t n_in, void* d_out, gr4hip_stream_t stream, const FirHooks& hk);

int gr4hip_fir_process(gr4hip_fir_t* f, const void* d_in, size_t n_in, void* d_out, size_t* n_out_p, gr4hip_stream_t stream) {
    GR4_REQUIRE(f, "fir_process: null handle");
    GR4_REQUIRE(n_in % f->decim == 0, "fir_process: n_in=%zu is not a multiple of decim=%zu", n_in, f->decim);
    const size_t n_out = n_in / f->decim;
    if (n_out_p) *n_out_p = n_out;
    if (n_in == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in && d_out, "fir_process: null device pointer");
    FirHooks hk;
    if (f->pre || f->post || f->folded != 1.0) { if (const int rc = fir_prepare_hooks(f, &hk, as_stream(stream))) return rc; }
    if (const int rc = fir_state_on(f, as_stream(stream))) return rc; // pending reset / tap upload / history rescale: onto this call's stream, in front of its launches
    // Neighbours that do not fold into the taps are load / store hooks of the register-window kernel.  Where a plain filter of this shape takes a matrix-pipe or
    // frequency-domain kernel instead (more than 32 taps on a long span, the float decimators), that kernel is worth more than the saved pass: add -> 256-tap FIR
    // measured 165 Gsamples/s hooked against 277 as an element-wise launch + the bf16 kernel.  There the program runs as ONE element-wise launch in front of
    // (behind) the filter's own -- same results, same history (the filter sees, and remembers, the prologue's output); everything else stays one launch.
    // (where the crossover lies, float, 2^27 samples: add -> 64 taps 432 hooked / 347 as two launches; -> 256 taps 165 / 258; the register-window kernel is FP32-bound
    // from ~48 taps per output on, the element-wise launch costs one pass at ~740 Gsamples/s)
    const size_t per_out    = ceil_div(f->ntaps, f->decim); // taps per computed output
    const bool   fast_shape = f->algo == GR4HIP_FIR_AUTO && ((f->decim == 1 && f->ntaps > (f->S == 2 ? (size_t)64 : (size_t)96) && f->ntaps <= (f->S == 2 ? (size_t)256 : (size_t)1024) && n_in >= kMfmaMinSamples) ||
                                                               (f->S == 1 && f->decim >= 2 && per_out > 12 && n_out >= ((size_t)1 << 14)));
    int        brc  = GR4HIP_OK;
    const bool band = hk.any && (fir_decim_f16_shape(f, n_in, d_in, d_out, true) || fir_decim_bf16_ready(f, n_in, d_in, d_out, &brc, as_stream(stream))); // (those kernels carry the hooks themselves)
    if (brc) return brc;
    auto around = [&]() -> int { // the programs as element-wise launches in front of / behind the plain filter
        hipStream_t st  = as_stream(stream);
        const void* src = d_in;
        if (hk.pre.n_ops > 0) {
            if (const int rc = f->d_pre.ensure(n_in * f->S * sizeof(float))) return rc;
            if (const int rc = ewise_run(hk.pre, f->dtype, d_in, f->d_pre.ptr, (long)n_in, st)) return rc;
            src = f->d_pre.ptr;
        }
        FirHooks none;
        if (const int rc = fir_process_core(f, src, n_in, d_out, stream, none)) return rc;
        if (hk.post.n_ops > 0) return ewise_run(hk.post, f->dtype, d_out, d_out, (long)n_out, st);
        return GR4HIP_OK;
    };
    if (hk.any && fast_shape && !band) return around();
    const int rc = fir_process_core(f, d_in, n_in, d_out, stream, hk);
    //  a hooked decimator whose phase rows do not fit the register-window kernel's LDS at any workgroup size (decimation ~100 with few taps) -- the one kernel
    // that carries hooks there -- is served like the fast shapes: nothing has been launched or moved when the core says UNSUPPORTED
    if (rc == GR4HIP_UNSUPPORTED && hk.any) return around();
    return rc;
}

} // extern "C"

static int fir_process_core(gr4hip_fir_t* f, const void* d_in, size_t n_in, void* d_out, gr4hip_stream_t stream, const FirHooks& hk) {
    const size_t n_out = n_in / f->decim;
    hipStream_t  st   = as_stream(stream);
    const float* x    = static_cast<const float*>(d_in);
    float*       y    = static_cast<float*>(d_out);
    const bool   plain = !hk.any;
    const int    algo  = plain ? f->algo : (int)GR4HIP_FIR_EXACT_F32; // (EXACT_F32 == "the register-window kernel only") // hooks run in the register-window kernel only: the matrix-pipe and frequency-domain kernels below are for plain filters
    const float* hist = (const float*)f->d_hist[f->cur].ptr;
    size_t       done = 0; // samples already produced by the frequency-domain path
    bool         mfma_wrote_hist = false;
    // a kernel that does not judge its own segments (the float32 matrix-pipe forms, the three-term bf16 direct forms) notes here where its part of the span starts and which
    // samples lie in front of it: fir_judge_kernel + fir_exact_kernel follow it at the end of the call (every kernel of this file answers to the same guard)
    size_t       unj_from = SIZE_MAX;
    const float* unj_hist = nullptr;
    // complex<float>, no decimation, <= 256 taps, long input: whole 8192-sample frames go through the fused FFT -> xH -> inverse
    // kernel (2 transforms per frame instead of 1024 flop per sample); the direct-form kernel finishes the remainder
    // (<= 96 taps: the direct form is faster -- write-bound 300 .. 356 Gsamples/s up to 64 taps, 311 .. 275 at 65 .. 96 taps on the bf16 matrix pipe, against
    // the fast convolution's 248 at every tap count, tools/cfir_taps_sweep.py -- and has no dynamic-range floor; 128 taps: 220, 256 taps: 151)
    // (round 4: the direct form on the f16 matrix pipe -- further down -- is as fast at 256 taps and faster below (129 taps 288 against 241, 192 taps 255 against 242,
    // 256 taps 237 against 240 Gsamples/s, tools/cfir_taps_sweep.py), its error is relative to the output, a non-finite sample reaches ntaps outputs instead of an
    // 8192-sample block and the call stays asynchronous: the fast convolution is left with the spans that kernel does not take -- unaligned ones)
    const bool f16c = f->S == 2 && f->decim == 1 && f->ntaps > 32 && f->ntaps <= 256 && n_in >= kMfmaMinSamples / 2 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(x)) & 15) == 0 &&
                      algo != GR4HIP_FIR_EXACT_F32 && !f->f32_user && !f->bf16_user && !dev_switch(kDevFirNoBf16x3) && !dev_switch(kDevFirNoF16x2) && f->hfKS >= 0 && plain;
    if (f->S == 2 && f->decim == 1 && f->ntaps > 96 && f->ntaps <= 256 && n_in >= kFdMinFrames * kFdFrame && algo == GR4HIP_FIR_AUTO && !f->fd_blocked && !f16c) {
        int rc = GR4HIP_OK;
        if (!f->fd) {
            rc = chain_fused_create(&f->fd, f->taps.data(), f->ntaps, kFdFrame, GR4HIP_WIN_NONE, GR4HIP_CHAIN_FUSED_FD);
            if (!rc && f->ntaps > 1 && f->guard_mode != GR4HIP_GUARD_OFF) chain_fused_set_measure(f->fd, true);
        }
        if (!rc) rc = f->d_hist256.ensure(256 * sizeof(float2));
        if (rc) return rc;
        hipLaunchKernelGGL(fir_hist_widen_kernel<float2>, dim3(1), dim3(256), 0, st, (const float2*)hist, (int)f->hcap, (float2*)f->d_hist256.ptr, 256);
        GR4_LAUNCH_CHECK();
        const size_t frames = n_in / kFdFrame;
        float        ratio;
        const bool   guarded = f->ntaps > 1 && f->guard_mode != GR4HIP_GUARD_OFF;
        if (guarded && f->guard_mode == GR4HIP_GUARD_STRICT) {
            // (round 5: nobody waits) the whole span on the fast convolution; every 8192-sample frame is judged behind it at the fast convolution's own threshold -- output power
            // below 0.04 x input power: its error floor, ~2e-6 of the INPUT rms, would show against such an output -- and the marked frames are evaluated again with float64
            // products (fir_judge_kernel + fir_exact_kernel on the same stream).  The measurement of an EARLIER launch, when it has arrived, moves a stream that rejects most of
            // its input to the direct form for good.
            if (chain_fused_power_ratio(f->fd, false, true, &ratio)) f->fd_ratio = ratio;
            f->fd_probed = true;
            if (f->fd_ratio >= 0.f && f->fd_ratio < 0.04f) f->fd_blocked = f->f32_products = true;
            else {
                const long nfd = (long)(frames * kFdFrame);
                rc = chain_fused_fir(f->fd, x, (const float*)f->d_hist256.ptr, frames, y, st);
                if (!rc) rc = f->d_flags_fd.ensure(frames);
                if (!rc) rc = fir_judge_launch(x, nfd, y, nfd, 1, 1, 13, 0.04f, (unsigned char*)f->d_flags_fd.ptr, st);
                if (!rc) rc = fir_exact_launch(x, nfd, hist, (int)f->hcap, (const float*)f->d_tapsf.ptr, (int)f->ntaps, 1, 1, y, nfd, (const unsigned char*)f->d_flags_fd.ptr, 13, nullptr, st);
                if (rc) return rc;
                done = (size_t)nfd;
                hist = x + (done - f->hcap) * 2;
            }
        } else {
        if (chain_fused_power_ratio(f->fd, false, true, &ratio)) f->fd_ratio = ratio; // a finished earlier launch: no waiting
        size_t probe = 0;
        if (!f->fd_probed && guarded) { // first fast convolution of this stream: eight frames, synchronously, before the span is committed to it
            probe = std::min<size_t>(frames, 8);
            rc    = chain_fused_fir(f->fd, x, (const float*)f->d_hist256.ptr, probe, y, st);
            if (rc) return rc;
            if (chain_fused_power_ratio(f->fd, true, true, &ratio)) f->fd_ratio = ratio;
            f->fd_probed = true;
        }
        if (guarded && f->fd_ratio >= 0.f && f->fd_ratio < 0.04f) {
            f->fd_blocked = f->f32_products = true; // the direct form below redoes the probed frames too (float32 products from here on)
        } else {
            if (probe < frames) {
                rc = chain_fused_fir(f->fd, x + probe * kFdFrame * 2, probe ? x + (probe * kFdFrame - 256) * 2 : (const float*)f->d_hist256.ptr, frames - probe, y + probe * kFdFrame * 2, st);
                if (rc) return rc;
            }
            done = frames * kFdFrame;
            hist = x + (done - f->hcap) * 2; // the hcap samples in front of the remainder are part of the input itself
        }
        }
    }
    // complex<float>, no decimation, 33..256 taps, whatever the fast convolution did not take (GR4HIP_FIR_TIME_DOMAIN, a stream the dynamic-range guard has
    // moved to the direct form, or both): the same block-Toeplitz product on the re and im planes of the interleaved samples (fir_mfma_c32_kernel)
    // (16-byte-aligned input too: the three-term bf16 form of the same product, fir_bf16.hip)
    // ... since round 4 on the f16 matrix pipe with two-term splits under a per-segment block exponent (fir_f16.hip): half the products, and every segment judges its
    // own output / input power -- so a stream the fast convolution's guard has handed over stays here (the segments that need float32 products get them inside the launch)
    if (f->S == 2 && f->decim == 1 && f->ntaps > 32 && f->ntaps <= 256 && n_in - done >= kMfmaMinSamples / 2 && ((reinterpret_cast<uintptr_t>(y + done * 2) | reinterpret_cast<uintptr_t>(x + done * 2)) & 15) == 0 &&
        algo != GR4HIP_FIR_EXACT_F32 && !f->f32_user && !f->bf16_user && !dev_switch(kDevFirNoBf16x3) && !dev_switch(kDevFirNoF16x2) && f->hfKS >= 0 && plain) {
        int rc = GR4HIP_OK;
        if (f->hfKS == 0) {
            std::vector<unsigned short> af;
            if (!fir_f16_make_afrag(f->taps.data(), f->ntaps, &f->hfKS, &af, 1, 0)) f->hfKS = -1;
            else {
                if (f->guard_ratio != kGuardSegmentRatio) { const float g = (float)(f->tap_power * f->guard_ratio); std::memcpy(af.data() + (size_t)f->hfKS * 1536 + 6, &g, 4); } // (the table's header, fir_f16_make_afrag: the tighter
                // threshold goes on the segment's WHOLE output power; the quietest-column statistic keeps its 21 dB -- at 15 dB it dips below on narrow-band noise alone: 1 % pass band, half of all segments marked)
                rc = f->d_hfrag.ensure(af.size() * sizeof(unsigned short));
                if (!rc) { hipError_t e = hipMemcpyAsync(f->d_hfrag.ptr, af.data(), af.size() * sizeof(unsigned short), hipMemcpyHostToDevice, st); if (e != hipSuccess) { set_error("fir: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
                if (rc) { f->hfKS = 0; return rc; }
            }
        }
        if (f->hfKS > 0) {
            float* nh = done == 0 ? (float*)f->d_hist[f->cur ^ 1].ptr : nullptr;
            const long nrem = (long)(n_in - done);
            rc = f->d_flags.ensure((size_t)ceil_div(nrem, 2048L));
            if (rc) return rc;
            rc = fir_f16_c32_launch(f->hfKS, x + done * 2, nrem, hist, (int)f->hcap, f->d_hfrag.ptr, y + done * 2, st, nh, f->guard_mode != GR4HIP_GUARD_OFF, (unsigned char*)f->d_flags.ptr);
            if (rc) return rc;
            rc = fir_exact_launch(x + done * 2, nrem, hist, (int)f->hcap, (const float*)f->d_tapsf.ptr, (int)f->ntaps, 1, 1, y + done * 2, nrem, (const unsigned char*)f->d_flags.ptr, 11, nullptr, st); // the marked segments again, on the FP64 matrix pipe
            if (rc) return rc;
            done = n_in;
            mfma_wrote_hist = nh != nullptr;
        }
    }
    // (round 5) complex, 257 .. 1792 taps: slices of 256 taps on the same kernel, each a pass over the input delayed by 256 p samples that adds to y (the float path's scheme below;
    // until then these filters took the f32 matrix pipe at its peak: 512 taps 36 Gsamples/s, 1024 taps 17).  The last slice judges the sums against the whole filter's threshold;
    // marked segments again with all the taps on the FP64 matrix pipe.
    if (f->S == 2 && f->decim == 1 && f->ntaps > 256 && f->ntaps <= 1792 /* (seven slices: a slice's window and its delay must fit the kernel's 2048-sample segment) */ && done == 0 && n_in >= kMfmaMinSamples / 2 && ((reinterpret_cast<uintptr_t>(y) | reinterpret_cast<uintptr_t>(x)) & 15) == 0 &&
        algo != GR4HIP_FIR_EXACT_F32 && !f->f32_user && !f->bf16_user && !dev_switch(kDevFirNoBf16x3) && !dev_switch(kDevFirNoF16x2) && f->hfKS >= 0 && plain) {
        int          rc = GR4HIP_OK;
        const size_t nslice = ceil_div(f->ntaps, (size_t)256);
        if (f->hfKS == 0) {
            std::vector<unsigned short> all;
            bool                        ok = true;
            f->hf_off.assign(nslice, 0);
            f->hf_ks.assign(nslice, 0);
            for (size_t p = 0; p < nslice && ok; ++p) {
                std::vector<unsigned short> af;
                const size_t                len = std::min<size_t>(256, f->ntaps - 256 * p);
                const int                   nat = std::max(3, (int)((len - 1 + 16 + 31) / 32));
                ok = fir_f16_make_afrag(f->taps.data() + 256 * p, len, &f->hf_ks[p], &af, 1, nat == 8 ? 9 : 0); // (the sliced kernel at 8 K-steps would keep two registers in scratch: such a slice runs the 9-step kernel on zero-padded taps)
                if (ok && p + 1 == nslice && f->guard_ratio != kGuardSegmentRatio) { const float g = (float)(f->tap_power * f->guard_ratio); std::memcpy(af.data() + (size_t)f->hf_ks[p] * 1536 + 6, &g, 4); } // (the judging slice carries the
                // library-internal caller's second verdict on the whole output power, like the one-pass table above)
                f->hf_off[p] = all.size();
                all.insert(all.end(), af.begin(), af.end());
            }
            if (!ok) f->hfKS = -1;
            else {
                rc = f->d_hfrag.ensure(all.size() * sizeof(unsigned short));
                if (!rc) { hipError_t e = hipMemcpyAsync(f->d_hfrag.ptr, all.data(), all.size() * sizeof(unsigned short), hipMemcpyHostToDevice, st); if (e != hipSuccess) { set_error("fir: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
                if (rc) return rc;
                f->hfKS = f->hf_ks[0];
            }
        }
        if (f->hfKS > 0) {
            float* nh = (float*)f->d_hist[f->cur ^ 1].ptr;
            rc = f->d_flags.ensure((size_t)ceil_div((long)n_in, 2048L));
            if (rc) return rc;
            const bool judged = f->guard_mode != GR4HIP_GUARD_OFF;
            for (size_t p = 0; p < nslice && !rc; ++p)
                rc = fir_f16_c32_launch(f->hf_ks[p], x, (long)n_in, hist, (int)f->hcap, (const unsigned short*)f->d_hfrag.ptr + f->hf_off[p], y, st, p == 0 ? nh : nullptr, p + 1 == nslice && judged,
                                        (unsigned char*)f->d_flags.ptr, (int)(256 * p), p > 0, (float)(f->tap_power * kGuardSegmentRatio));
            if (rc) return rc;
            rc = fir_exact_launch(x, (long)n_in, hist, (int)f->hcap, (const float*)f->d_tapsf.ptr, (int)f->ntaps, 1, 1, y, (long)n_in, (const unsigned char*)f->d_flags.ptr, 11, nullptr, st);
            if (rc) return rc;
            done = n_in;
            mfma_wrote_hist = true;
        }
    }
    static const size_t kCBfMinTaps = [] { const char* e = std::getenv("GR4HIP_CFIR_BF16_MIN_TAPS"); return e ? (size_t)std::atoi(e) : (size_t)33; }(); // developer knob: 65 puts 33..64 taps back on the f32 MFMA (measured 5-8 % slower: 299-316 against 322-333 Gsamples/s); below 33 taps the register-window kernel is ahead up to 27 taps and within 3 % from there (tools/cfir_bf16_threshold.py)
    if (done < n_in && f->S == 2 && f->decim == 1 && f->ntaps >= kCBfMinTaps && f->ntaps <= 256 && n_in - done >= kMfmaMinSamples / 2 && ((reinterpret_cast<uintptr_t>(y + done * 2) | reinterpret_cast<uintptr_t>(x + done * 2)) & 15) == 0 &&
        !no_bf16x3(f) && plain) {
        int rc = GR4HIP_OK;
        if (f->bfKS == 0) {
            std::vector<unsigned short> af;
            fir_bf16_make_afrag(f->taps.data(), f->ntaps, &f->bfKS, &af, 1, 0);
            rc = f->d_bfrag.ensure(af.size() * sizeof(unsigned short));
            if (!rc) { hipError_t e = hipMemcpyAsync(f->d_bfrag.ptr, af.data(), af.size() * sizeof(unsigned short), hipMemcpyHostToDevice, st); if (e != hipSuccess) { set_error("fir: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
            if (rc) { f->bfKS = 0; return rc; }
        }
        float* nh = done == 0 ? (float*)f->d_hist[f->cur ^ 1].ptr : nullptr;
        rc = fir_bf16_c32_launch(f->bfKS, x + done * 2, (long)(n_in - done), hist, (int)f->hcap, f->d_bfrag.ptr, y + done * 2, st, nh);
        if (rc) return rc;
        unj_from = done; unj_hist = hist;
        done = n_in;
        mfma_wrote_hist = nh != nullptr;
    }
    const bool exact = algo == GR4HIP_FIR_EXACT_F32; // only the register-window kernel: its products are the taps' (padded to a multiple of 4 per phase), nothing wider
    if (!exact && done < n_in && f->S == 2 && f->decim == 1 && f->ntaps > 32 && f->ntaps <= 256 && n_in - done >= kMfmaMinSamples / 2 && (reinterpret_cast<uintptr_t>(y + done * 2) & 15) == 0) {
        int rc = GR4HIP_OK;
        if (f->mKS == 0) {
            std::vector<float> af;
            fir_mfma_make_afrag(f->taps.data(), f->ntaps, 1, &f->mKp, &f->mKS, &af);
            rc = f->d_afrag.ensure(af.size() * sizeof(float));
            if (!rc) { hipError_t e = hipMemcpyAsync(f->d_afrag.ptr, af.data(), af.size() * sizeof(float), hipMemcpyHostToDevice, st); if (e != hipSuccess) { set_error("fir: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
            if (rc) { f->mKS = 0; return rc; }
        }
        const float* hk = hist; // (hcap = bit_ceil(ntaps) is Kp for 33 .. 256 taps: read in place; a span served by this launch alone also gets its next history from it)
        if ((int)f->hcap != f->mKp) {
            rc = f->d_histc.ensure(256 * sizeof(float2));
            if (rc) return rc;
            hipLaunchKernelGGL(fir_hist_widen_kernel<float2>, dim3(1), dim3(256), 0, st, (const float2*)hist, (int)f->hcap, (float2*)f->d_histc.ptr, f->mKp);
            GR4_LAUNCH_CHECK();
            hk = (const float*)f->d_histc.ptr;
        }
        float* nh = (done == 0 && (int)f->hcap == f->mKp) ? (float*)f->d_hist[f->cur ^ 1].ptr : nullptr;
        rc = fir_mfma_c32_launch(f->mKS, x + done * 2, (long)(n_in - done), hk, (const float*)f->d_afrag.ptr, y + done * 2, st, nh);
        if (rc) return rc;
        unj_from = done; unj_hist = hist;
        done = n_in;
        mfma_wrote_hist = nh != nullptr;
    }
    // float, no decimation, 33 .. 256 taps, long 16-byte-aligned span: the same product on the bf16 matrix pipe with three-term splits of samples and taps
    // (fir_bf16.hip: float32 accuracy, 2.4 times less matrix-pipe time than the f32 MFMA -- HBM-bound instead of MFMA-bound)
    // (384 .. 1024 taps: slices of 256 taps, each a pass over the input delayed by 256 p samples that adds to y: 512 taps 115 instead of 90 Gsamples/s on the
    // register-window kernel, 1024 taps 50.5 instead of 47; below 384 and above 1024 taps the extra passes cost more than they save -- measured)
    if (f->S == 1 && f->decim == 1 && f->ntaps > 32 && f->ntaps <= 3840 /* (fifteen slices: a slice's window and its delay must fit the kernel's 4096-sample segment; round 5's last day: 257 .. 383 and 1025 .. 3840 taps as slices too -- the three-term bf16 slices they were measured against in round 3 are not what they compete with any more: 257 / 300 / 383 taps 167 / 148 / 119 Gsamples/s on the other kernels, 1100 taps 44) */ && n_in >= kMfmaMinSamples && ((reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(d_in)) & 15) == 0 &&
        algo == GR4HIP_FIR_AUTO && !no_bf16x3(f) && !f->bf16_user && !dev_switch(kDevFirNoF16x2) && f->hfKS >= 0 && plain) {
        // ... on the f16 matrix pipe with two-term splits under a per-segment block exponent (fir_f16.hip): three products per tap instead of six
        int          rc = GR4HIP_OK;
        const size_t nslice = ceil_div(f->ntaps, (size_t)256);
        if (f->hfKS == 0) {
            std::vector<unsigned short> all;
            bool                        ok = true;
            f->hf_off.assign(nslice, 0);
            f->hf_ks.assign(nslice, 0);
            for (size_t p = 0; p < nslice && ok; ++p) {
                std::vector<unsigned short> af;
                const size_t                len = std::min<size_t>(256, f->ntaps - 256 * p);
                ok = fir_f16_make_afrag(f->taps.data() + 256 * p, len, &f->hf_ks[p], &af, 1, 0);
                f->hf_off[p] = all.size();
                all.insert(all.end(), af.begin(), af.end());
            }
            if (!ok) f->hfKS = -1;
            else {
                rc = f->d_hfrag.ensure(all.size() * sizeof(unsigned short));
                if (!rc) { hipError_t e = hipMemcpyAsync(f->d_hfrag.ptr, all.data(), all.size() * sizeof(unsigned short), hipMemcpyHostToDevice, st); if (e != hipSuccess) { set_error("fir: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
                if (rc) return rc;
                f->hfKS = f->hf_ks[0];
            }
        }
        if (f->hfKS > 0) {
            float* nh = (float*)f->d_hist[f->cur ^ 1].ptr;
            rc = f->d_flags.ensure(ceil_div(n_in, (size_t)4096));
            if (rc) return rc;
            // every segment judges its own output / input power (fir_f16.hip); of a long filter's slices the LAST one does, on the sums it leaves in y (the whole filter's
            // outputs) against the whole filter's threshold.  Marked segments -- rejected, or with samples the block exponent cannot carry in ANY slice -- are evaluated
            // again with all the taps on the FP64 matrix pipe (fir_exact.hip)
            const double h2 = f->tap_power;
            for (size_t p = 0; p < nslice && !rc; ++p)
                rc = fir_f16_launch(f->hf_ks[p], x, (long)n_in, hist, (int)f->hcap, (const unsigned short*)f->d_hfrag.ptr + f->hf_off[p], y, st, p == 0 ? nh : nullptr, 0, 0, 1, (int)(256 * p), p > 0,
                                    p + 1 == nslice && f->guard_mode != GR4HIP_GUARD_OFF, (unsigned char*)f->d_flags.ptr, 0, (nslice > 1 || f->guard_ratio != kGuardSegmentRatio) ? (float)(h2 * f->guard_ratio) : 0.f);
            if (rc) return rc;
            rc = fir_exact_launch(x, (long)n_in, hist, (int)f->hcap, (const float*)f->d_tapsf.ptr, (int)f->ntaps, 1, 0, y, (long)n_in, (const unsigned char*)f->d_flags.ptr, 12, nullptr, st);
            if (rc) return rc;
            done = n_in;
            mfma_wrote_hist = true;
        }
    }
    if (done == 0 && f->S == 1 && f->decim == 1 && f->ntaps > 32 && (f->ntaps <= 256 || (f->ntaps >= 384 && f->ntaps <= 1024)) && n_in >= kMfmaMinSamples && ((reinterpret_cast<uintptr_t>(d_out) | reinterpret_cast<uintptr_t>(d_in)) & 15) == 0 &&
        algo == GR4HIP_FIR_AUTO && !no_bf16x3(f) && plain) {
        int          rc = GR4HIP_OK;
        const size_t nslice = ceil_div(f->ntaps, (size_t)256);
        if (f->bfKS == 0) {
            std::vector<unsigned short> all;
            f->bf_off.assign(nslice, 0);
            f->bf_ks.assign(nslice, 0);
            for (size_t p = 0; p < nslice; ++p) {
                std::vector<unsigned short> af;
                const size_t                len = std::min<size_t>(256, f->ntaps - 256 * p);
                fir_bf16_make_afrag(f->taps.data() + 256 * p, len, &f->bf_ks[p], &af, 1, 0);
                f->bf_off[p] = all.size();
                all.insert(all.end(), af.begin(), af.end());
            }
            rc = f->d_bfrag.ensure(all.size() * sizeof(unsigned short));
            if (!rc) { hipError_t e = hipMemcpyAsync(f->d_bfrag.ptr, all.data(), all.size() * sizeof(unsigned short), hipMemcpyHostToDevice, st); if (e != hipSuccess) { set_error("fir: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
            if (rc) return rc;
            f->bfKS = f->bf_ks[0];
        }
        float* nh = (float*)f->d_hist[f->cur ^ 1].ptr;
        for (size_t p = 0; p < nslice && !rc; ++p)
            rc = fir_bf16_launch(f->bf_ks[p], x, (long)n_in, hist, (int)f->hcap, (const unsigned short*)f->d_bfrag.ptr + f->bf_off[p], y, st, p == 0 ? nh : nullptr, 0, 0, 1, (int)(256 * p), p > 0);
        if (rc) return rc;
        unj_from = 0; unj_hist = hist;
        done = n_in;
        mfma_wrote_hist = true;
    }
    // float, no decimation, 33..256 taps, long 16-byte-aligned output: block-Toeplitz product on the f32 MFMA units (about twice the
    // rate of the register-window VALU kernel, which is FP32-issue bound from ~48 taps on)
    if (!exact && done == 0 && f->S == 1 && f->decim == 1 && f->ntaps > 32 && f->ntaps <= 256 && n_in >= kMfmaMinSamples && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0) {
        int rc = GR4HIP_OK;
        if (f->mKS == 0) {
            std::vector<float> af;
            fir_mfma_make_afrag(f->taps.data(), f->ntaps, 1, &f->mKp, &f->mKS, &af);
            rc = f->d_afrag.ensure(af.size() * sizeof(float));
            if (!rc) { hipError_t e = hipMemcpyAsync(f->d_afrag.ptr, af.data(), af.size() * sizeof(float), hipMemcpyHostToDevice, st); if (e != hipSuccess) { set_error("fir: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
            if (rc) { f->mKS = 0; return rc; }
        }
        // hcap = bit_ceil(ntaps) IS Kp for 33 .. 256 taps: the block's history is read as it lies, and the launch writes the next one (one launch per call
        // instead of three: what a scheduler's 64 Ki-sample chunks cost is launches)
        const float* hk = hist;
        if ((int)f->hcap != f->mKp) {
            rc = f->d_hist256.ensure(256 * sizeof(float2));
            if (rc) return rc;
            hipLaunchKernelGGL(fir_hist_widen_kernel<float>, dim3(1), dim3(256), 0, st, hist, (int)f->hcap, (float*)f->d_hist256.ptr, f->mKp);
            GR4_LAUNCH_CHECK();
            hk = (const float*)f->d_hist256.ptr;
        }
        float* nh = (int)f->hcap == f->mKp ? (float*)f->d_hist[f->cur ^ 1].ptr : nullptr;
        rc = fir_mfma_launch(f->mKS, x, (long)n_in, hk, (const float*)f->d_afrag.ptr, y, (long)((n_in + 3) & ~(size_t)3), (long)n_in, 1, st, nh);
        if (rc) return rc;
        unj_from = 0; unj_hist = hist;
        done = n_in;
        mfma_wrote_hist = nh != nullptr;
    }
    // float, decimate by 8, 97 .. 1025 taps (BASELINE configs[2]), long 16-byte-aligned span: the band form on the f16 matrix pipe, two-term splits under a per-segment block
    // exponent, the K-steps split over the four waves, every segment judged and the rejected ones evaluated again with float32 products inside the launch (fir_decim_f16.hip)
    // -- its error is relative to the output, so it needs no host-side guard and the call stays asynchronous
    if (done == 0 && fir_decim_f16_shape(f, n_in, d_in, d_out, hk.any)) { // (hooked calls too: the kernel carries the programs, fir_exact_kernel<true> runs them again on marked segments)
        int rc = GR4HIP_OK;
        if (f->dhKQ == 0) {
            std::vector<unsigned short> tab;
            if (!fir_decim_f16_make_table(f->taps.data(), f->ntaps, f->decim, &f->dhKQ, &tab, f->S == 2)) f->dhKQ = -1;
            else {
                rc = f->d_dhtab.ensure(tab.size() * sizeof(unsigned short));
                if (!rc) { hipError_t e = hipMemcpyAsync(f->d_dhtab.ptr, tab.data(), tab.size() * sizeof(unsigned short), hipMemcpyHostToDevice, st); if (e != hipSuccess) { set_error("fir: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
                if (rc) { f->dhKQ = 0; return rc; }
            }
        }
        if (f->dhKQ > 0) {
            const long segf = 8192 / (long)f->decim; // the kernel's segment: 8192 input floats = this many output floats
            rc = f->d_flags.ensure((size_t)ceil_div((long)(n_out * f->S), segf));
            if (rc) return rc;
            rc = fir_decim_f16_launch((int)f->decim, f->dhKQ, x, (long)(n_in * f->S), hist, (int)(f->hcap * f->S), f->d_dhtab.ptr, y, (long)(n_out * f->S), st, (float*)f->d_hist[f->cur ^ 1].ptr,
                                      f->guard_mode != GR4HIP_GUARD_OFF, (unsigned char*)f->d_flags.ptr, f->S == 2, &hk.pre, &hk.post);
            if (rc) return rc;
            // the marked segments again on the FP64 matrix pipe (fir_exact.hip; in samples: a complex segment is half as many outputs)
            rc = fir_exact_launch(x, (long)n_in, hist, (int)f->hcap, (const float*)f->d_tapsf.ptr, (int)f->ntaps, (int)f->decim, f->S == 2, y, (long)n_out, (const unsigned char*)f->d_flags.ptr,
                                  ilog2((size_t)(segf / f->S)), nullptr, st, 1, 0, 0, 0, 0, &hk.pre, &hk.post);
            if (rc) return rc;
            done = n_in;
            mfma_wrote_hist = true;
        }
    }
    // float, decimate by 2 .. 12 with a window of <= 1152 samples (taps - 1 + 15 D), long 16-byte-aligned span: the band form with three-term bf16 products
    // (fir_bf16.hip; windows beyond 288 samples with the K-steps split over the four waves)
    // complex<float> (real taps), decimate by 3 .. 16: the same kernels on the interleaved stream read as floats (fir_decim_bf16_make_afrag: the rows of a tile
    // alternate between the re and im phases of the window) -- the register-window kernel below stages a decimator's samples one by one (decimate by 8, 64 taps:
    // 171 G input samples/s against 483 here; by 16: 97 against 445; by 2 it is ahead: 389 / 307 against 348 / 252 at 16 / 64 taps)
    // These kernels take the neighbours' programs themselves (load hook where the samples are split into bf16 planes, store hook on the output tile).
    if (done == 0) {
        int rc = GR4HIP_OK;
        if (fir_decim_bf16_ready(f, n_in, d_in, d_out, &rc, st)) {
            float* nh = (float*)f->d_hist[f->cur ^ 1].ptr;
            // every segment judges its own output / input power (the three-term products' error is relative to the products); the marked ones are evaluated again on the
            // FP64 matrix pipe behind the launch (fir_exact.hip), which runs the launch's load / store programs too
            const bool judged = f->guard_mode != GR4HIP_GUARD_OFF && f->ntaps > 1;
            int        seg_out = 0;
            if (judged) { rc = f->d_flags.ensure((size_t)ceil_div((long)(n_out * f->S), 512L)); if (rc) return rc; }
            const double h2 = f->tap_power;
            rc = fir_decim_bf16_launch(f->bdKS, (int)f->decim, f->bdHb, x, (long)(n_in * f->S), hist, (int)(f->hcap * f->S), f->d_bdfrag.ptr, y, (long)(n_out * f->S), st, nh,
                                       hk.pre.n_ops > 0 ? &hk.pre : nullptr, hk.post.n_ops > 0 ? &hk.post : nullptr, f->S == 2, judged ? (unsigned char*)f->d_flags.ptr : nullptr,
                                       (float)(h2 * f->guard_ratio), &seg_out);
            if (rc == GR4HIP_OK && judged)
                rc = fir_exact_launch(x, (long)n_in, hist, (int)f->hcap, (const float*)f->d_tapsf.ptr, (int)f->ntaps, (int)f->decim, f->S == 2, y, (long)n_out, (const unsigned char*)f->d_flags.ptr,
                                      ilog2((size_t)(seg_out / f->S)), nullptr, st, 1, 0, 0, 0, 0, &hk.pre, &hk.post);
            if (rc == GR4HIP_OK) { done = n_in; mfma_wrote_hist = true; }
            else if (rc == GR4HIP_UNSUPPORTED) f->bdKS = -1; // (the staged segment does not fit: do not ask again)
            else return rc;
        } else if (rc) return rc;
    }
    // float, decimate by 8, <= 1024 taps, long 16-byte-aligned span: overlap-save blocks of 8192 samples in the frequency domain (~35 lane-operations per
    // input sample instead of 2 K / 8 flop: HBM / power-bound instead of FP32-bound); a partial last block rides in the same launch
    constexpr size_t kDfHopS = 7168, kDfMinBlocks = 64;
    if (done == 0 && f->S == 1 && algo == GR4HIP_FIR_AUTO && !f->fd_blocked && fir_decim_fd_supported(f->ntaps, f->decim) && f->ntaps <= 1024 && n_in >= kDfMinBlocks * kDfHopS &&
        (reinterpret_cast<uintptr_t>(d_in) & 15) == 0 && !dev_switch(kDevFirNoDecimFd)) {
        int rc = GR4HIP_OK;
        if (!f->dfd) rc = fir_decim_fd_create(&f->dfd, f->taps.data(), f->ntaps);
        if (rc) return rc;
        // whole blocks and the span's partial last block in ONE launch (+ one small launch in front that widens the history and stages the partial block):
        // sending the < 7168 leftover samples through the polyphase kernel cost a 44 us single-workgroup launch behind a 164 us transform kernel
        // the same dynamic-range guard as the complex fast convolution above: the error floor of the transforms is ~2e-6 of the INPUT rms, and an anti-alias
        // filter is exactly where strong out-of-band power is removed.  Strict: this launch's own measurement decides before the call returns (the polyphase
        // kernels below redo the span); deferred: a finished earlier launch decides; from the first rejection on the stream stays on the polyphase kernels
        const bool guarded = f->guard_mode != GR4HIP_GUARD_OFF && f->ntaps > 1;
        float      ratio;
        if (guarded && fir_decim_fd_power_ratio(f->dfd, false, &ratio)) f->fd_ratio = ratio;
        // threshold: this kernel's floor is one forward 4096-point and one inverse 1024-point transform -- measured (tools/decim_fd_floor.py) max 2.5e-7 .. 3.7e-7, rms 7e-8 of
        // the INPUT rms per output sample, six times below the fused chain's -- so 1e-5 of the OUTPUT rms holds down to a power ratio of (4e-7 / 1e-5)^2 = 1.6e-3; 2.5e-3
        // (-26 dB) with margin.  White noise through a DC-gain-1 low-pass of cut-off fc passes 2 fc of its power: every anti-alias filter down to fc = 0.00125 stays here.
        constexpr float kDecimFdMinPowerRatio = 2.5e-3f;
        if (guarded && f->fd_ratio >= 0.f && f->fd_ratio < kDecimFdMinPowerRatio) f->fd_blocked = f->f32_products = true; // (an earlier launch's measurement, read without waiting)
        if (!f->fd_blocked) {
            rc = fir_decim_fd_run(f->dfd, x, n_in, hist, (int)f->hcap, y, st, guarded);
            if (rc) return rc;
            done = n_in;
            if (guarded && f->guard_mode == GR4HIP_GUARD_STRICT) { // (round 5: nobody waits) segments of 2048 outputs judged behind the launch at this kernel's threshold, the marked ones again in float64
                rc = f->d_flags_fd.ensure((size_t)ceil_div((long)n_out, 2048L));
                if (!rc) rc = fir_judge_launch(x, (long)n_in, y, (long)n_out, (int)f->decim, 0, 11, kDecimFdMinPowerRatio, (unsigned char*)f->d_flags_fd.ptr, st);
                if (!rc) rc = fir_exact_launch(x, (long)n_in, hist, (int)f->hcap, (const float*)f->d_tapsf.ptr, (int)f->ntaps, (int)f->decim, 0, y, (long)n_out, (const unsigned char*)f->d_flags_fd.ptr, 11, nullptr, st);
                if (rc && rc != GR4HIP_UNSUPPORTED) return rc;
            }
        }
    }
    // float, decimate by kBandMinDecim .. 128, long span: the band form of the contraction (samples in stream order, the decimation in the A operand)
    static const size_t kBandMinDecim = [] { const char* e = std::getenv("GR4HIP_FIR_BAND_MIN_DECIM"); return e ? (size_t)std::atoi(e) : (size_t)10; }(); // (developer switch: where the band form takes over from the polyphase form)
    if (done == 0 && f->S == 1 && f->decim >= kBandMinDecim && f->decim <= 128 && n_out >= (1u << 14) && algo == GR4HIP_FIR_AUTO) {
        int rc = GR4HIP_OK;
        if (f->bandKp == 0) {
            std::vector<float> row;
            fir_decim_band_make_row(f->taps.data(), f->ntaps, f->decim, &f->bandKp, &row);
            rc = f->d_band.ensure(row.size() * sizeof(float));
            if (!rc) { hipError_t e = hipMemcpyAsync(f->d_band.ptr, row.data(), row.size() * sizeof(float), hipMemcpyHostToDevice, st); if (e != hipSuccess) { set_error("fir: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
            if (rc) { f->bandKp = 0; return rc; }
        }
        rc = fir_decim_band_launch((int)f->decim, f->bandKp, x, hist, (int)f->hcap, (const float*)f->d_band.ptr, y, (long)n_out, (long)n_in, st);
        if (rc == GR4HIP_OK) { done = n_in; unj_from = 0; unj_hist = hist; }
        else if (rc != GR4HIP_UNSUPPORTED) return rc; // UNSUPPORTED: the tile does not fit the LDS, keep the kernels below
    }
    // float polyphase decimator with >= 16 taps per phase and a long span: the same contraction with the D phase products summed in
    // one accumulator tile (BASELINE configs[2]: decim 8, 1024 taps)
    if (!exact && done == 0 && f->S == 1 && f->decim >= 2 && ceil_div(f->ntaps, f->decim) >= 16 && ceil_div(f->ntaps, f->decim) <= 256 && n_out >= (1u << 14) &&
        (reinterpret_cast<uintptr_t>(d_out) & 15) == 0) {
        int rc = GR4HIP_OK;
        if (f->mKS == 0) {
            std::vector<float> af;
            fir_mfma_make_afrag_decim(f->taps.data(), f->ntaps, f->decim, &f->mKp, &f->mKS, &af);
            rc = f->d_afrag.ensure(af.size() * sizeof(float));
            if (!rc) { hipError_t e = hipMemcpyAsync(f->d_afrag.ptr, af.data(), af.size() * sizeof(float), hipMemcpyHostToDevice, st); if (e != hipSuccess) { set_error("fir: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
            if (rc) { f->mKS = 0; return rc; }
        }
        const int hl = f->mKp * (int)f->decim; // history the kernel wants: Kp D samples
        rc = f->d_hist256.ensure((size_t)hl * sizeof(float));
        if (rc) return rc;
        const int hu = (int)std::min<size_t>(f->hcap, (size_t)hl); // newest hu samples of the block's history are what the kernel can see
        hipLaunchKernelGGL(fir_hist_widen_kernel<float>, dim3((unsigned)ceil_div(hl, 256)), dim3(256), 0, st, hist + (f->hcap - hu), hu, (float*)f->d_hist256.ptr, hl);
        GR4_LAUNCH_CHECK();
        rc = fir_mfma_decim_launch(f->mKS, (int)f->decim, x, (const float*)f->d_hist256.ptr, (const float*)f->d_afrag.ptr, y, (long)n_out, st);
        if (rc == GR4HIP_OK) { done = n_in; unj_from = 0; unj_hist = hist; }
        else if (rc != GR4HIP_UNSUPPORTED) return rc; // UNSUPPORTED: the de-interleaved segment does not fit the LDS, keep the VALU kernel
    }
    const int E  = 4 / f->S;
    int       rc = done == n_in ? GR4HIP_OK : GR4HIP_UNSUPPORTED;
    bool      hist_written = mfma_wrote_hist;
    for (int bs : {256, 128, 64}) {
        if (done == n_in) break;
        const size_t Lf  = (size_t)bs * kFirR + 4 * (size_t)f->G;
        const size_t lds = f->decim * (Lf + (size_t)f->G * E) * sizeof(float);
        if (lds > 150 * 1024) continue;
        const float* xr = x + done * f->S;
        float*       yr = y + (done / f->decim) * f->S;
        const long   ni = (long)(n_in - done), no = (long)((n_in - done) / f->decim);
        float*       nh = done == 0 ? (float*)f->d_hist[f->cur ^ 1].ptr : nullptr; // the whole span in this launch: it writes the next history itself
        if (f->S == 1) rc = bs == 256 ? fir_launch<1, 256>(f, xr, hist, yr, ni, no, lds, st, nh, hk) : bs == 128 ? fir_launch<1, 128>(f, xr, hist, yr, ni, no, lds, st, nh, hk) : fir_launch<1, 64>(f, xr, hist, yr, ni, no, lds, st, nh, hk);
        else rc = bs == 256 ? fir_launch<2, 256>(f, xr, hist, yr, ni, no, lds, st, nh, hk) : bs == 128 ? fir_launch<2, 128>(f, xr, hist, yr, ni, no, lds, st, nh, hk) : fir_launch<2, 64>(f, xr, hist, yr, ni, no, lds, st, nh, hk);
        hist_written = nh != nullptr && rc == GR4HIP_OK;
        break;
    }
    if (rc == GR4HIP_UNSUPPORTED && done == 0 && f->S == 1 && f->decim >= 2 && plain) { // D phase rows do not fit the LDS at any workgroup size: the band form at any span length
        if (f->bandKp == 0) {
            std::vector<float> row;
            fir_decim_band_make_row(f->taps.data(), f->ntaps, f->decim, &f->bandKp, &row);
            int rb = f->d_band.ensure(row.size() * sizeof(float));
            if (!rb) { hipError_t e = hipMemcpyAsync(f->d_band.ptr, row.data(), row.size() * sizeof(float), hipMemcpyHostToDevice, st); if (e != hipSuccess) { set_error("fir: upload failed: %s", hipGetErrorString(e)); rb = GR4HIP_RUNTIME_ERROR; } }
            if (rb) { f->bandKp = 0; return rb; }
        }
        rc = fir_decim_band_launch((int)f->decim, f->bandKp, x, hist, (int)f->hcap, (const float*)f->d_band.ptr, y, (long)n_out, (long)n_in, st);
        if (rc == GR4HIP_OK) { unj_from = 0; unj_hist = hist; }
    }
    if (rc == GR4HIP_UNSUPPORTED && done == 0 && plain) { // no tiling fits: the one-output-per-lane kernel (the guard's second evaluation follows it where that one fits)
        const unsigned g = (unsigned)ceil_div((long)n_out, 256L);
        if (f->S == 1) hipLaunchKernelGGL(fir_generic_kernel<1>, dim3(g), dim3(256), 0, st, x, hist, (int)f->hcap, (const float*)f->d_tapsf.ptr, (int)f->ntaps, (long)f->decim, y, (long)n_out, (long)n_in);
        else hipLaunchKernelGGL(fir_generic_kernel<2>, dim3(g), dim3(256), 0, st, x, hist, (int)f->hcap, (const float*)f->d_tapsf.ptr, (int)f->ntaps, (long)f->decim, y, (long)n_out, (long)n_in);
        GR4_LAUNCH_CHECK();
        rc = GR4HIP_OK; unj_from = 0; unj_hist = hist;
    }
    if (rc == GR4HIP_UNSUPPORTED) { set_error("fir_process: ntaps=%zu decim=%zu does not fit the LDS tiling", f->ntaps, f->decim); return rc; }
    if (rc) return rc;
    if (unj_from != SIZE_MAX && f->guard_mode != GR4HIP_GUARD_OFF && f->ntaps > 1) {
        const long ni = (long)(n_in - unj_from), no = ni / (long)f->decim;
        const int  shift = f->S == 2 ? 10 : 11;
        const float* xr = x + unj_from * f->S;
        float*       yr = y + (unj_from / f->decim) * f->S;
        if (no > 0) {
            rc = f->d_flags.ensure((size_t)ceil_div(no, 1L << shift));
            if (rc) return rc;
            rc = fir_judge_launch(xr, ni, yr, no, (int)f->decim, f->S == 2, shift, (float)(f->tap_power * kGuardSegmentRatio), (unsigned char*)f->d_flags.ptr, st,
                                  f->guard_ratio != kGuardSegmentRatio ? (float)(f->tap_power * f->guard_ratio) : 0.f); // (a tighter ratio: on the segment's whole output power -- the quietest quarter dips on narrow-band noise)
            if (rc) return rc;
            rc = fir_exact_launch(xr, ni, unj_hist, (int)f->hcap, (const float*)f->d_tapsf.ptr, (int)f->ntaps, (int)f->decim, f->S == 2, yr, no, (const unsigned char*)f->d_flags.ptr, shift, nullptr, st);
            if (rc && rc != GR4HIP_UNSUPPORTED) return rc; // (UNSUPPORTED: decimation x taps beyond what the second evaluation stages -- the float32 sums stand)
        }
    }
    if (!hist_written) {
        const long tot = (long)f->hcap * f->S;
        hipLaunchKernelGGL(fir_hist_update_kernel, dim3((unsigned)ceil_div(tot, 256L)), dim3(256), 0, st, static_cast<const float*>(d_in),
                           (const float*)f->d_hist[f->cur].ptr, (float*)f->d_hist[f->cur ^ 1].ptr, (long)n_in, (int)f->hcap, f->S);
        GR4_LAUNCH_CHECK();
    }
    f->cur ^= 1;
    f->pos += (long)n_in;
    return GR4HIP_OK;
}

// ------------------------------------------------------------------------------------------------ decimating FIR -> IIR cascade (BASELINE configs[2])
int  gr4hip_internal_iir_fusable(gr4hip_iir_t* f, const float** d_tab, int* nsec, const float** d_state_in, float** d_state_out, int* warm_blocks, hipStream_t st); // iir.hip
void gr4hip_internal_iir_commit(gr4hip_iir_t* f);

extern "C" int gr4hip_fir_iir_process(gr4hip_fir_t* f, gr4hip_iir_t* iir, const float* d_in, size_t n_in, float* d_out, size_t* n_out_p, int mode, gr4hip_stream_t stream) {
    GR4_REQUIRE(f && iir, "fir_iir_process: null handle");
    GR4_REQUIRE(mode >= GR4HIP_FIR_IIR_AUTO && mode <= GR4HIP_FIR_IIR_TWO_LAUNCHES, "fir_iir_process: unknown mode %d", mode);
    GR4_REQUIRE(f->dtype == GR4HIP_F32, "fir_iir_process: a float filter in front of the float cascade");
    GR4_REQUIRE(n_in % f->decim == 0, "fir_iir_process: n_in=%zu is not a multiple of decim=%zu", n_in, f->decim);
    const size_t n_out = n_in / f->decim;
    if (n_out_p) *n_out_p = n_out;
    if (n_in == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in && d_out, "fir_iir_process: null device pointer");
    hipStream_t  st = as_stream(stream);
    const auto two_launches = [&](const float* x, size_t n, float* y) -> int { // the decimated stream through HBM (f->d_pre as scratch)
        if (const int rc = f->d_mid.ensure(std::max<size_t>(n / f->decim, 1) * sizeof(float))) return rc;
        float* mid = static_cast<float*>(f->d_mid.ptr);
        if (const int rc = gr4hip_fir_process(f, x, n, mid, nullptr, stream)) return rc;
        return gr4hip_iir_process(iir, mid, n / f->decim, y, stream);
    };
    constexpr size_t kHop = 7168, kMinBlocks = 64;
    const float*     tab = nullptr;
    const float*     st_in = nullptr;
    float*           st_out = nullptr;
    int              nsec = 0, warm = 0;
    const size_t     whole = n_in / kHop * kHop;
    if (f->folded != 1.0 && !f->pre && !f->post) { // (a gain prologue / epilogue that was taken off again left its factor in the taps on this path: gr4hip_fir_process undoes it
                                                   // through fir_prepare_hooks, the fused launch below never came by there)
        FirHooks hk0;
        if (const int rc = fir_prepare_hooks(f, &hk0, st)) return rc;
    }
    if (const int rc = fir_state_on(f, st)) return rc;
    const bool       fusable = mode == GR4HIP_FIR_IIR_ONE_LAUNCH && f->S == 1 && f->decim == 8 && f->algo == GR4HIP_FIR_AUTO && !f->fd_blocked && !f->pre && !f->post && fir_decim_fd_supported(f->ntaps, f->decim) && f->ntaps <= 1024 &&
                         whole >= kMinBlocks * kHop && (reinterpret_cast<uintptr_t>(d_in) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 7) == 0 && !dev_switch(kDevFirNoDecimFd) &&
                         gr4hip_internal_iir_fusable(iir, &tab, &nsec, &st_in, &st_out, &warm, st);
    if (!fusable) return two_launches(d_in, n_in, d_out);
    int rc = GR4HIP_OK;
    if (!f->dfd) rc = fir_decim_fd_create(&f->dfd, f->taps.data(), f->ntaps);
    if (rc) return rc;
    // the decimator's dynamic-range guard as in gr4hip_fir_process: its own measurement decides (strict: before this call returns -- nothing of the launch is kept, the
    // cascade's state included), a finished earlier launch decides (deferred)
    constexpr float kDecimFdMinPowerRatio = 2.5e-3f;
    const bool      guarded = f->guard_mode != GR4HIP_GUARD_OFF && f->ntaps > 1;
    float           ratio;
    if (guarded && fir_decim_fd_power_ratio(f->dfd, false, &ratio)) f->fd_ratio = ratio;
    if (guarded && f->guard_mode == GR4HIP_GUARD_DEFERRED && f->fd_ratio >= 0.f && f->fd_ratio < kDecimFdMinPowerRatio) { f->fd_blocked = f->f32_products = true; return two_launches(d_in, n_in, d_out); }
    const float*     hist = (const float*)f->d_hist[f->cur].ptr;
    const DecimFdIir fuse{tab, nsec, st_in, st_out, warm};
    rc = fir_decim_fd_run(f->dfd, d_in, whole, hist, (int)f->hcap, d_out, st, guarded, &fuse);
    if (rc) return rc;
    if (guarded && f->guard_mode == GR4HIP_GUARD_STRICT) {
        if (fir_decim_fd_power_ratio(f->dfd, true, &ratio)) f->fd_ratio = ratio;
        if (f->fd_ratio >= 0.f && f->fd_ratio < kDecimFdMinPowerRatio) { f->fd_blocked = f->f32_products = true; return two_launches(d_in, n_in, d_out); } // (neither handle has moved)
    }
    gr4hip_internal_iir_commit(iir);
    {   // the filter's history moves behind the whole blocks
        const long tot = (long)f->hcap;
        hipLaunchKernelGGL(fir_hist_update_kernel, dim3((unsigned)ceil_div(tot, 256L)), dim3(256), 0, st, d_in, hist, (float*)f->d_hist[f->cur ^ 1].ptr, (long)whole, (int)f->hcap, 1);
        GR4_LAUNCH_CHECK();
        f->cur ^= 1;
        f->pos += (long)whole;
    }
    if (whole < n_in) return two_launches(d_in + whole, n_in - whole, d_out + whole / 8); // what is left of the span (< one block): the polyphase kernel and the cascade's own
    return GR4HIP_OK;
}

// (library-internal, chain.hip) make `d_last256` -- the 256 complex samples in front of the next input sample -- this filter's history: the fused chain
// hands its stream over to the direct-form kernels when the dynamic-range guard switches algorithms
// (library-internal, chain.hip) the per-segment guard's threshold as a multiple of (sum b^2): before the first call (the matrix-pipe kernels' tables carry it).  Non-decimating filters only
int gr4hip_internal_fir_set_guard_ratio(gr4hip_fir_t* f, double ratio) {
    GR4_REQUIRE(f && f->decim == 1 && ratio > 0 && ratio <= 1.0, "fir_set_guard_ratio: a non-decimating filter and a ratio in (0, 1] expected");
    f->guard_ratio = ratio;
    if (f->hfKS > 0) f->hfKS = 0; // (a table built already is built again with the new threshold)
    return GR4HIP_OK;
}
int gr4hip_internal_fir_load_history(gr4hip_fir_t* f, const float* d_last256, hipStream_t st) {
    GR4_REQUIRE(f && f->S == 2 && f->hcap <= 256, "fir_load_history: complex filter with <= 256 samples of history expected");
    f->zero_hist    = false; // (what was pending for the history is superseded; the taps go up with the next call)
    f->hist_rescale = 1.0;
    GR4_HIP_TRY(hipMemcpyAsync(f->d_hist[f->cur].ptr, d_last256 + (256 - f->hcap) * 2, f->hcap * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
    return GR4HIP_OK;
}

extern "C" {

int gr4hip_fir_destroy(gr4hip_fir_t* f) {
    if (f && f->fd) chain_fused_destroy(f->fd);
    if (f && f->dfd) fir_decim_fd_destroy(f->dfd);
    delete f;
    return GR4HIP_OK;
}

int gr4hip_decimate(int dtype, const void* d_in, size_t n_in, size_t decim, void* d_out, size_t* n_out_p, gr4hip_stream_t stream) {
    const size_t es = dtype_size(dtype);
    GR4_REQUIRE(es, "decimate: unknown dtype %d", dtype);
    GR4_REQUIRE(decim >= 1, "decimate: decim must be >= 1");
    const size_t n_out = ceil_div(n_in, decim); // i % decim == 0 for i in [0, n_in)
    if (n_out_p) *n_out_p = n_out;
    if (n_out == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in && d_out, "decimate: null device pointer");
    const unsigned grid = (unsigned)std::min<size_t>(ceil_div(n_out, (size_t)256), 4096);
    hipStream_t    st   = as_stream(stream);
    switch (es) {
    case 1: hipLaunchKernelGGL(decimate_kernel<uint8_t>, dim3(grid), dim3(256), 0, st, (const uint8_t*)d_in, (uint8_t*)d_out, (long)n_out, (long)decim); break;
    case 2: hipLaunchKernelGGL(decimate_kernel<uint16_t>, dim3(grid), dim3(256), 0, st, (const uint16_t*)d_in, (uint16_t*)d_out, (long)n_out, (long)decim); break;
    case 4: hipLaunchKernelGGL(decimate_kernel<uint32_t>, dim3(grid), dim3(256), 0, st, (const uint32_t*)d_in, (uint32_t*)d_out, (long)n_out, (long)decim); break;
    case 8: hipLaunchKernelGGL(decimate_kernel<uint64_t>, dim3(grid), dim3(256), 0, st, (const uint64_t*)d_in, (uint64_t*)d_out, (long)n_out, (long)decim); break;
    default: hipLaunchKernelGGL(decimate_kernel<ulonglong2>, dim3(grid), dim3(256), 0, st, (const ulonglong2*)d_in, (ulonglong2*)d_out, (long)n_out, (long)decim); break;
    }
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

} // extern "C"
