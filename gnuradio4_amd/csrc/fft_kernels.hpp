// fft_kernels.hpp -- device code of the FFT block shared by fft.hip and fft_fast_pk.hip: small DFTs, the output epilogue and the
// compile-time-plan kernel fft_fast_kernel<log2 N> with its launcher.  (Two translation units because hipcc's SLP vectoriser --
// v_pk_add/mul/fma_f32 on register pairs it has to assemble with v_mov -- costs 7-13 % at N = 1024 / 4096 and GAINS 15-17 % at
// N = 256 / 8192: the library is built with -fno-slp-vectorize except fft_fast_pk.hip, which instantiates those two sizes.)
#pragma once
#include "common.hpp"
#include "buffer_ops.hpp"
#include "fft_radix.hpp"
#include "ewise.hpp"

#include <cfloat>
#include <cmath>
#include <complex>


namespace gr4 {

struct FftPlanDev {
    int N;
    int npass;
    int radix[16];
    int tf;  // threads per frame
    int fpb; // frames per block
    int smooth; // 1: mixed-radix plan of fft_smooth.hpp (radices 2 .. 16, N not a power of two)
};

struct FftOutputs {
    float* spectrum;  // [frames][N][2]
    float* re;        // complex in: [frames][N]; real in: [frames][N/2] (bins N/2..N-1, fft.hpp:221-227)
    float* im;
    float* mag;       // shifted (complex) / first half (real)
    float* phase;     // final phase when !unwrap
    float* phase_raw; // natural-order raw atan2 (only when unwrap; finished by unwrap_kernel)
    float* mag2;      // natural order
    float* ranges;    // fast kernels only: [frames][4][2] {min, max} of magnitude, phase, Re, Im (fft.hpp:229-232); null = not requested
    int    in_db, in_deg, real_input;
    EwiseHook mag2_post; // per-sample float blocks behind a PowerSpectrum (MultiplyConst / AddConst ...: gr4hip_fft_set_epilogue): applied to |X|^2 before its store
};

// every requested output of one natural-order bin k of one frame (fft.hpp:164-166, fft_common.hpp:20-56, 91-123)
__device__ __forceinline__ void emit_bin(const FftOutputs& out, long frame, int N, int k, float2 X) {
    const int half = N / 2;
    const int nout = out.real_input ? half : N;
    if (out.spectrum) reinterpret_cast<float2*>(out.spectrum)[frame * N + k] = X;
    if (out.mag2) {
        float m = fmaf(X.x, X.x, X.y * X.y);
        if (out.mag2_post.n_ops > 0) m = ewise_hook1<float>(m, out.mag2_post, frame * N + k);
        out.mag2[frame * N + k] = m;
    }
    if (out.real_input) {
        if (k >= half) {
            if (out.re) out.re[frame * half + (k - half)] = X.x;
            if (out.im) out.im[frame * half + (k - half)] = X.y;
        }
    } else {
        if (out.re) out.re[frame * N + k] = X.x;
        if (out.im) out.im[frame * N + k] = X.y;
    }
    if (out.real_input && k >= half) return; // computeHalfSpectrum: first N/2 bins, never rotated
    const int ko = out.real_input ? k : (k + N - half) % N; // shiftSpectrum = std::rotate by N/2: bin N/2 comes first (odd N on the Bluestein path)
    if (out.mag) {
        float m = hypotf(X.x, X.y) * 2.f / (float)N;
        if (out.in_db) m = (m > 0.f) ? 20.f * log10f(m) : -FLT_MAX;
        out.mag[frame * nout + ko] = m;
    }
    if (out.phase || out.phase_raw) {
        float ph = atan2f(X.y, X.x);
        if (out.phase_raw) {
            out.phase_raw[frame * nout + k] = ph;
        } else {
            if (out.in_deg) ph = ph * 180.f * 0.318309886183790671538f;
            out.phase[frame * nout + ko] = ph;
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Fast path for N = 256 ... 8192: compile-time Stockham plan 16 x 16 x R3 (x 2 for 8192), 16 points per lane, 512 lanes =
// 512 / (N/16) frames per workgroup iteration, persistent workgroups (2 per CU).  The first pass reads its 16 points per lane
// straight from global memory (window fused), the last pass emits every output straight from registers (lane t holds bins
// t + j N/16: coalesced), so a frame makes 2 (N <= 4096: 16 x 16 x R3) or 3 LDS round trips instead of log8(N) + 1, all of them
// through a buffer padded by one float2 per 32 (the stride-16 scatter of the first pass is conflict-free).  Twiddles: one or two
// table values per lane and pass live in registers, the powers b^r come from two interleaved chains (depth <= 7).

// REAL = true: a frame of 2N REAL samples is transformed as N complex points z[n] = x[2n] + i x[2n+1] (the float array read as float2,
// window pairs likewise) and split afterwards:  X[k] = (Z[k] + conj Z[N-k]) / 2 + W_2N^k (Z[k] - conj Z[N-k]) / (2i),  k = 0 .. N
// -- half the butterflies of the "imaginary part = 0" form.  tw is then the table of W_2N^j (every second entry is W_N^j), and the
// outputs follow the real-input conventions of the block: magnitude / phase = bins 0..N-1, Re / Im = bins N..2N-1 = conj of bins N..1.
template <int LOG2N, bool REAL = false>
__global__ __launch_bounds__(512, 4) void fft_fast_kernel(const float* __restrict__ in, const float* __restrict__ window, const float2* __restrict__ tw,
                                                          FftOutputs out, long n_frames) {
    constexpr int N = 1 << LOG2N, T = N / 16, FPB = 512 / T, NP = N + N / 32, HALF = N / 2;
    constexpr int TWS = REAL ? 2 : 1; // stride of W_N^j in the table
    static_assert(!REAL || LOG2N <= 12, "the real-input split is implemented for the 16 x 16 x R3 plans");
    constexpr int R3  = N >= 4096 ? 16 : N / 256; // 256 -> 1 (no third pass)
    constexpr int R4  = N / (256 * R3);           // 8192 -> 2
    constexpr int B3  = R3 > 1 ? 16 / R3 : 1;     // third-pass butterflies per lane
    constexpr int NB3 = N / R3;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    auto P = [](int i) { return i + (i >> 5); };
    const int t0 = threadIdx.x % T;

    // per-lane twiddle bases, exact table values (tw[j] = W_N^j)
    const float2 w2a_ = tw[TWS * (t0 & 15) * (N / 256)], w2b_ = tw[TWS * 2 * (t0 & 15) * (N / 256)]; // W_256^k, W_256^2k
    // third pass, butterfly b of the lane: k = t + b T (< 256), W_{256 R3}^k = W_{256 R3}^t W_16^b  (256 R3 = N below 8192)
    float2 w3_ = make_float2(1.f, 0.f), w3sq_ = make_float2(1.f, 0.f);
    // (8192: the lane's third-pass butterfly is i = (t >> 1) + 256 (t & 1), k = t >> 1, see below)
    constexpr int kShift3 = R4 == 2 ? 1 : 0;
    if constexpr (R3 > 1) w3_ = tw[TWS * ((t0 >> kShift3) & 255) * (N / (256 * R3))];
    if constexpr (R3 == 16) w3sq_ = tw[TWS * 2 * ((t0 >> kShift3) & 255) * (N / 4096)];
    float2 w4_ = make_float2(1.f, 0.f);
    if constexpr (R4 == 2) w4_ = tw[t0 >> 1]; // W_8192^{t >> 1}
    const float2 wr_  = REAL ? tw[t0] : make_float2(1.f, 0.f); // REAL: W_2N^t, the split twiddle of bin t + j T is this times W_32^j
    const rsrc_t rwin = make_rsrc(window, window ? (REAL ? 2 : 1) * N * 4u : 0u);

    const long ngroups = (n_frames + FPB - 1) / FPB;
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        // all global accesses of this iteration go through buffer descriptors over the group's FPB frames: one 32-bit lane offset +
        // compile-time bin offsets, frames past n_frames fall out of range (loads return 0, stores are dropped)
        // (the lane id is laundered every iteration: lane-dependent LDS / buffer offsets are recomputed instead of being hoisted out of
        // the loop as dozens of loop-invariant VGPRs)
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid));
        // same for the twiddle bases: their power chains are loop-invariant and hipcc would keep all ~35 products in registers
        float2 w2a = w2a_, w2b = w2b_, w3sq = w3sq_, w4 = w4_;
        asm volatile("" : "+v"(w2a.x), "+v"(w2a.y), "+v"(w2b.x), "+v"(w2b.y), "+v"(w3sq.x), "+v"(w3sq.y), "+v"(w4.x), "+v"(w4.y));
        float2 w3 = w3_;
        asm volatile("" : "+v"(w3.x), "+v"(w3.y));
        const int      fl    = tid / T, t = tid % T;
        float2*        buf   = lds + fl * NP;
        const long     f0    = g * FPB;
        const unsigned nlive = (unsigned)(n_frames - f0 < FPB ? n_frames - f0 : FPB);
        const int      vN    = fl * N + t; // element offset of (frame slot, bin t) in an [FPB][N] group (input side)
        float2         v[16];
        // ---- pass 1 (p = 1, radix 16) from global memory; window (fft.hpp:148-162); real input becomes (x*w, 0)
        if constexpr (REAL) {
            const rsrc_t rx = make_rsrc(in + f0 * N * 2, nlive * N * 8u); // 2N floats per frame = N float2
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = buf_load_f2(rx, vN * 8, r * T * 8);
            if (window) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float2 w = buf_load_f2(rwin, t * 8, r * T * 8); v[r].x *= w.x; v[r].y *= w.y; }
            }
        } else if (out.real_input) {
            const rsrc_t rx = make_rsrc(in + f0 * N, nlive * N * 4u);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = make_float2(buf_load_f(rx, vN * 4, r * T * 4), 0.f);
        } else {
            const rsrc_t rx = make_rsrc(in + f0 * N * 2, nlive * N * 8u);
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = buf_load_f2(rx, vN * 8, r * T * 8);
        }
        if (!REAL && window) {
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float w = buf_load_f(rwin, t * 4, r * T * 4); v[r].x *= w; v[r].y *= w; }
        }
        fft16<1>(v);
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[P(16 * t + r)] = v[perm16(r)];
        __syncthreads();
        // ---- pass 2 (p = 16, radix 16): butterfly i = t, k = t & 15
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = buf[P(t + r * T)];
        __syncthreads();
        apply_powers(v, w2a, w2b);
        fft16<1>(v);
        float2 X[16]; // X[j] = bin t + j T
        if constexpr (R3 == 1) { // N = 256: bins t + 16 q
#pragma unroll
            for (int q = 0; q < 16; ++q) X[q] = v[perm16(q)];
        } else {
#pragma unroll
            for (int q = 0; q < 16; ++q) buf[P((t & ~15) * 16 + (t & 15) + 16 * q)] = v[perm16(q)];
            __syncthreads();
            // ---- pass 3 (p = 256, radix R3): butterflies i_b = t + b T, k = i_b & 255
            // 8192: the two butterflies whose outputs meet in the final radix-2 step (i and i + 256) go to NEIGHBOURING lanes
            const int i3 = R4 == 2 ? (t >> 1) + 256 * (t & 1) : t;
#pragma unroll
            for (int b = 0; b < B3; ++b)
#pragma unroll
                for (int r = 0; r < R3; ++r) v[b * R3 + r] = buf[P(i3 + b * T + r * NB3)];
            __syncthreads();
            if constexpr (R3 == 16) {
                apply_powers(v, w3, w3sq);
                fft16<1>(v);
            } else {
#pragma unroll
                for (int b = 0; b < B3; ++b) {
                    const float2 wb = b == 0 ? w3 : cmul(w3, w32(2 * b));
                    float2       pw = wb;
#pragma unroll
                    for (int r = 1; r < R3; ++r) {
                        v[b * R3 + r] = cmul(v[b * R3 + r], pw);
                        if (r + 1 < R3) pw = cmul(pw, wb);
                    }
                    dft_small<R3>(v + b * R3);
                }
            }
            if constexpr (R4 == 1) { // last pass: k = i_b, bins i_b + 256 q = t + (b + q B3) T
#pragma unroll
                for (int b = 0; b < B3; ++b)
#pragma unroll
                    for (int q = 0; q < R3; ++q) X[b + q * B3] = v[b * R3 + (R3 == 16 ? perm16(q) : q)];
            } else { // N = 8192, pass 4 (p = 4096, radix 2) across the lane pair through DPP, no fourth LDS round trip:
                // even lane a = X3[i][q], odd lane b = X3[i + 256][q];  bin k3 + 256 q = a + W b (even lane), + 4096: a - W b (odd lane),
                // W = W_8192^{k3 + 256 q} = W_8192^{k3} W_32^q
                const int   par = t & 1;
                const float sgn = par ? -1.f : 1.f;
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float2       u  = v[perm16(q)];
                    const float2 uw = cmul(u, q == 0 ? w4 : cmul(w4, w32(q)));
                    u.x = par ? uw.x : u.x;
                    u.y = par ? uw.y : u.y;
                    const float2 o = make_float2(lane_xor1(u.x), lane_xor1(u.y));
                    X[q] = make_float2(fmaf(sgn, u.x, o.x), fmaf(sgn, u.y, o.y));
                }
            }
        }
        // per-frame {min, max} of the four DataSet signals, accumulated while the values are in registers (no second pass over HBM)
        float rmin[4], rmax[4];
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) { rmin[g4] = FLT_MAX; rmax[g4] = -FLT_MAX; }
        auto track = [&](int sig, float v) { rmin[sig] = fminf(rmin[sig], v); rmax[sig] = fmaxf(rmax[sig], v); };
        if constexpr (REAL) {
            // ---- split: X[j] holds Z[k], k = t + j T; the partner Z[(N - k) mod N] lives in another lane -> one more LDS round trip
            float2 wr = wr_;
            asm volatile("" : "+v"(wr.x), "+v"(wr.y));
#pragma unroll
            for (int j = 0; j < 16; ++j) buf[P(t + j * T)] = X[j];
            __syncthreads();
            float2 Zp[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) Zp[j] = buf[P((N - (t + j * T)) & (N - 1))];
            __syncthreads(); // buf is rewritten by the next iteration
            const float nyq = X[0].x - X[0].y; // bin N (lane t = 0 only): Re Z[0] - Im Z[0]
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const float2 e  = make_float2(0.5f * (X[j].x + Zp[j].x), 0.5f * (X[j].y - Zp[j].y));
                const float2 o  = make_float2(0.5f * (X[j].y + Zp[j].y), -0.5f * (X[j].x - Zp[j].x)); // -i (Z - conj Zp) / 2
                const float2 wk = j == 0 ? wr : cmul(wr, w32(j));
                X[j]            = cadd(e, cmul(wk, o));
            }
            const int vH = fl * N + t; // outputs are [frames][N]
            if (out.mag) {
                const rsrc_t r = make_rsrc(out.mag + f0 * N, nlive * N * 4u);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float m = hypotf(X[j].x, X[j].y) * 2.f / (float)(2 * N);
                    if (out.in_db) m = (m > 0.f) ? 20.f * log10f(m) : -FLT_MAX;
                    buf_store_f(r, m, vH * 4, j * T * 4);
                    track(0, m);
                }
            }
            if (out.phase || out.phase_raw) {
                const rsrc_t r = make_rsrc((out.phase_raw ? out.phase_raw : out.phase) + f0 * N, nlive * N * 4u);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float ph = atan2f(X[j].y, X[j].x);
                    if (!out.phase_raw && out.in_deg) ph = ph * 180.f * 0.318309886183790671538f;
                    buf_store_f(r, ph, vH * 4, j * T * 4);
                    if (!out.phase_raw) track(1, ph);
                }
            }
            // Re / Im: output index m <-> bin N + m = conj(bin N - m); the lane holding bin k = t + j T (k >= 1) owns m = N - k, bin N itself is nyq
            if (out.re) {
                const rsrc_t r = make_rsrc(out.re + f0 * N, nlive * N * 4u);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const bool  dc = j == 0 && t == 0;
                    const float v_ = dc ? nyq : X[j].x;
                    buf_store_f(r, v_, (fl * N + (dc ? 0 : N - t - j * T)) * 4, 0); // (no negative scalar offset: the range check sees the lane offset only)
                    track(2, v_);
                }
            }
            if (out.im) {
                const rsrc_t r = make_rsrc(out.im + f0 * N, nlive * N * 4u);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const bool  dc = j == 0 && t == 0;
                    const float v_ = dc ? 0.f : -X[j].y;
                    buf_store_f(r, v_, (fl * N + (dc ? 0 : N - t - j * T)) * 4, 0);
                    track(3, v_);
                }
            }
        } else {
        // ---- every requested output, straight from registers (fft.hpp:164-166, fft_common.hpp:20-56, 91-123)
        // X[j] is bin eN + j ST of the lane's frame; eS + soff(j) is the same bin after fftshift
        constexpr int ST = R4 == 2 ? 256 : T;
        const int     eN = R4 == 2 ? (t >> 1) + 4096 * (t & 1) : fl * N + t;
        const int     eS = R4 == 2 ? (t >> 1) + 4096 * (1 - (t & 1)) : eN;
        auto          soff = [](int j) { return R4 == 2 ? j * ST : (j * ST + HALF) % N; };
        if (out.spectrum) {
            const rsrc_t r = make_rsrc(out.spectrum + f0 * N * 2, nlive * N * 8u);
#pragma unroll
            for (int j = 0; j < 16; ++j) buf_store_f2(r, X[j], eN * 8, j * ST * 8);
        }
        if (out.mag2) {
            const rsrc_t r = make_rsrc(out.mag2 + f0 * N, nlive * N * 4u);
            float m[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) m[j] = fmaf(X[j].x, X[j].x, X[j].y * X[j].y);
            if (out.mag2_post.n_ops > 0) ewise_apply<float, 16>(m, out.mag2_post.ops, out.mag2_post.n_ops, out.mag2_post.has_div, [](int) { return 0L; }); // (float programs have no index-dependent op)
#pragma unroll
            for (int j = 0; j < 16; ++j) buf_store_f(r, m[j], eN * 4, j * ST * 4);
        }
        if (!out.real_input) { // N bins; magnitude and phase are fftshift-ed (bin k -> (k + N/2) mod N), Re/Im natural
            if (out.re) {
                const rsrc_t r = make_rsrc(out.re + f0 * N, nlive * N * 4u);
#pragma unroll
                for (int j = 0; j < 16; ++j) { buf_store_f(r, X[j].x, eN * 4, j * ST * 4); track(2, X[j].x); }
            }
            if (out.im) {
                const rsrc_t r = make_rsrc(out.im + f0 * N, nlive * N * 4u);
#pragma unroll
                for (int j = 0; j < 16; ++j) { buf_store_f(r, X[j].y, eN * 4, j * ST * 4); track(3, X[j].y); }
            }
            if (out.mag) {
                const rsrc_t r = make_rsrc(out.mag + f0 * N, nlive * N * 4u);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float m = hypotf(X[j].x, X[j].y) * 2.f / (float)N;
                    if (out.in_db) m = (m > 0.f) ? 20.f * log10f(m) : -FLT_MAX;
                    buf_store_f(r, m, eS * 4, soff(j) * 4);
                    track(0, m);
                }
            }
            if (out.phase || out.phase_raw) {
                const rsrc_t r = make_rsrc((out.phase_raw ? out.phase_raw : out.phase) + f0 * N, nlive * N * 4u);
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    float ph = atan2f(X[j].y, X[j].x);
                    if (out.phase_raw) {
                        buf_store_f(r, ph, eN * 4, j * ST * 4); // natural order, finished by unwrap_kernel
                    } else {
                        if (out.in_deg) ph = ph * 180.f * 0.318309886183790671538f;
                        buf_store_f(r, ph, eS * 4, soff(j) * 4);
                        track(1, ph);
                    }
                }
            }
        } else { // real input: Re/Im = bins N/2..N-1 (fft.hpp:221-227), magnitude / phase = bins 0..N/2-1, never rotated
            // below 8192 the lane holds bins t + j T: upper half = j >= 8; at 8192 the odd lanes hold the upper half
            const int  eH  = R4 == 2 ? (t >> 1) : fl * HALF + t;
            const bool hi  = R4 == 2 ? (t & 1) != 0 : true, lo = R4 == 2 ? (t & 1) == 0 : true;
            constexpr int J0 = R4 == 2 ? 0 : 8, J1 = R4 == 2 ? 16 : 8; // [J0, 16) upper-half slots, [0, J1) lower-half slots
            if (out.re && hi) {
                const rsrc_t r = make_rsrc(out.re + f0 * HALF, nlive * HALF * 4u);
#pragma unroll
                for (int j = J0; j < 16; ++j) { buf_store_f(r, X[j].x, eH * 4, (j - J0) * ST * 4); track(2, X[j].x); }
            }
            if (out.im && hi) {
                const rsrc_t r = make_rsrc(out.im + f0 * HALF, nlive * HALF * 4u);
#pragma unroll
                for (int j = J0; j < 16; ++j) { buf_store_f(r, X[j].y, eH * 4, (j - J0) * ST * 4); track(3, X[j].y); }
            }
            if (out.mag && lo) {
                const rsrc_t r = make_rsrc(out.mag + f0 * HALF, nlive * HALF * 4u);
#pragma unroll
                for (int j = 0; j < J1; ++j) {
                    float m = hypotf(X[j].x, X[j].y) * 2.f / (float)N;
                    if (out.in_db) m = (m > 0.f) ? 20.f * log10f(m) : -FLT_MAX;
                    buf_store_f(r, m, eH * 4, j * ST * 4);
                    track(0, m);
                }
            }
            if ((out.phase || out.phase_raw) && lo) {
                const rsrc_t r = make_rsrc((out.phase_raw ? out.phase_raw : out.phase) + f0 * HALF, nlive * HALF * 4u);
#pragma unroll
                for (int j = 0; j < J1; ++j) {
                    float ph = atan2f(X[j].y, X[j].x);
                    if (!out.phase_raw && out.in_deg) ph = ph * 180.f * 0.318309886183790671538f;
                    buf_store_f(r, ph, eH * 4, j * ST * 4);
                    if (!out.phase_raw) track(1, ph);
                }
            }
        }
        } // !REAL
        if (out.ranges) { // reduce over the T lanes of the frame: butterflies inside the wave, then (T > 64) through LDS
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
#pragma unroll
                for (int off = (T < 64 ? T : 64) / 2; off > 0; off >>= 1) {
                    rmin[g4] = fminf(rmin[g4], __shfl_xor(rmin[g4], off));
                    rmax[g4] = fmaxf(rmax[g4], __shfl_xor(rmax[g4], off));
                }
            }
            if constexpr (T > 64) {
                __syncthreads(); // everybody is done with buf (the last LDS reads were before the previous barrier, but waves may lag)
                float* red = reinterpret_cast<float*>(buf);
                if ((tid & 63) == 0) {
#pragma unroll
                    for (int g4 = 0; g4 < 4; ++g4) { red[(t >> 6) * 8 + g4] = rmin[g4]; red[(t >> 6) * 8 + 4 + g4] = rmax[g4]; }
                }
                __syncthreads();
                if (t == 0) {
                    for (int w = 1; w < T / 64; ++w)
#pragma unroll
                        for (int g4 = 0; g4 < 4; ++g4) { rmin[g4] = fminf(rmin[g4], red[w * 8 + g4]); rmax[g4] = fmaxf(rmax[g4], red[w * 8 + 4 + g4]); }
                }
                __syncthreads(); // red lives in buf: the next iteration writes there
            }
            if (t == 0 && f0 + fl < n_frames) {
                const bool have[4] = {out.mag != nullptr, out.phase != nullptr, out.re != nullptr, out.im != nullptr};
#pragma unroll
                for (int g4 = 0; g4 < 4; ++g4) {
                    out.ranges[((f0 + fl) * 4 + g4) * 2 + 0] = have[g4] ? rmin[g4] : 0.f;
                    out.ranges[((f0 + fl) * 4 + g4) * 2 + 1] = have[g4] ? rmax[g4] : 0.f;
                }
            }
        }
    }
}

template <int LOG2N, bool REAL = false>
static int fft_fast_launch(const float* d_in, const float* d_window, const float2* d_tw, const FftOutputs& o, long n_frames, hipStream_t st) {
    constexpr int    N = 1 << LOG2N, FPB = 512 / (N / 16);
    constexpr size_t lds = (size_t)FPB * (N + N / 32) * sizeof(float2);
    static PerDevice per_device; // the > 64 KiB LDS opt-in is per device
    bool             first = false;
    int              dev = -1;
    const int        n_cu = per_device.current(&first, &dev);
    GR4_REQUIRE(n_cu != 0, "fft: cannot query the current device");
    if (first) {
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fft_fast_kernel<LOG2N, REAL>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        per_device.done(dev, -n_cu);
    }
    const long groups = (n_frames + FPB - 1) / FPB;
    // one workgroup per group of frames, not a persistent grid: measured +8 ... 13 % at N <= 1024, +7 % at 8192 (the dispatcher refills a CU as soon as
    // a workgroup retires, and the per-workgroup set-up is two twiddle loads per lane); the frame loop in the kernel only matters beyond 2^31 groups
    const long grid   = groups < 0x7fffffffL ? groups : 0x7fffffffL;
    hipLaunchKernelGGL((fft_fast_kernel<LOG2N, REAL>), dim3((unsigned)grid), dim3(512), lds, st, d_in, d_window, d_tw, o, n_frames);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}


// the two sizes whose kernels are compiled with the SLP vectoriser on (fft_fast_pk.hip)
int fft_fast_launch_256(const float* d_in, const float* d_window, const float2* d_tw, const FftOutputs& o, long n_frames, hipStream_t st);
int fft_fast_launch_8192(const float* d_in, const float* d_window, const float2* d_tw, const FftOutputs& o, long n_frames, hipStream_t st);

} // namespace gr4
