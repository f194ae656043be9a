// fft_smooth.hpp -- the FFT block at {2,3,5}-smooth sizes that are not powers of two (N = 1000, 1536, 3000, ...; 6 <= N <= 8192): mixed-radix Stockham
// passes in LDS, one kernel launch.  SimdFFT takes exactly these sizes on its fast path (algorithm/.../fourier/SimdFFT.hpp:348-375 canProcessSize, radix order
// :398); before this kernel they went through the chirp (Bluestein) convolution at 3 transforms of twice the size.
//
// A frame is N / tf-th shared by tf lanes; a pass of radix R has N / R butterflies and every lane holds up to 16 / R of them (<= 16 points) in registers between
// the two barriers of the in-place autosort step (read all, barrier, write all).  Radices 2 .. 16: 2, 4, 8, 16 and 3, 5, 9 as hand-written butterflies,
// 6 = 2 x 3, 10 = 2 x 5, 12 = 4 x 3, 15 = 3 x 5 through the prime-factor index map (Good-Thomas: coprime factors need no inner twiddles).  Pass twiddles
// W_{pR}^{r k}: two exact table values per butterfly (W^k, W^{2k}) and two interleaved power chains of depth <= 7, as in the power-of-two kernels.
// The workgroup is persistent (frames g, g + gridDim, ...), loads and stores are natural order and coalesced, every output of the block comes from emit_bin.
#pragma once
#include "fft_kernels.hpp"

namespace gr4 {

// ---- prime butterflies (forward, W = e^{-2 pi i / R}), natural order in and out
__device__ __forceinline__ void dft3(float2& x0, float2& x1, float2& x2) {
    constexpr float c = 0.86602540378443864676f; // sin(2 pi / 3)
    const float2 s = caddf(x1, x2), d = csubf(x1, x2);
    const float2 m = make_float2(fmaf(-0.5f, s.x, x0.x), fmaf(-0.5f, s.y, x0.y));
    x0 = caddf(x0, s);
    x1 = make_float2(fmaf(c, d.y, m.x), fmaf(-c, d.x, m.y)); // m - i c d
    x2 = make_float2(fmaf(-c, d.y, m.x), fmaf(c, d.x, m.y)); // m + i c d
}
__device__ __forceinline__ void dft5(float2& x0, float2& x1, float2& x2, float2& x3, float2& x4) {
    constexpr float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f; // cos(2 pi / 5), cos(4 pi / 5)
    constexpr float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;  // sin(2 pi / 5), sin(4 pi / 5)
    const float2 a1 = caddf(x1, x4), a2 = caddf(x2, x3), b1 = csubf(x1, x4), b2 = csubf(x2, x3);
    const float2 m1 = make_float2(fmaf(c1, a1.x, fmaf(c2, a2.x, x0.x)), fmaf(c1, a1.y, fmaf(c2, a2.y, x0.y)));
    const float2 m2 = make_float2(fmaf(c2, a1.x, fmaf(c1, a2.x, x0.x)), fmaf(c2, a1.y, fmaf(c1, a2.y, x0.y)));
    const float2 n1 = make_float2(fmaf(s1, b1.x, s2 * b2.x), fmaf(s1, b1.y, s2 * b2.y));
    const float2 n2 = make_float2(fmaf(s2, b1.x, -s1 * b2.x), fmaf(s2, b1.y, -s1 * b2.y));
    x0 = caddf(x0, caddf(a1, a2));
    x1 = make_float2(m1.x + n1.y, m1.y - n1.x); // m1 - i n1
    x4 = make_float2(m1.x - n1.y, m1.y + n1.x);
    x2 = make_float2(m2.x + n2.y, m2.y - n2.x);
    x3 = make_float2(m2.x - n2.y, m2.y + n2.x);
}
template <int R>
__device__ __forceinline__ void dft_prime_pow(float2* v); // natural-order DFT of v[0 .. R) for R in {2, 3, 4, 5, 8, 9, 16}
template <> __device__ __forceinline__ void dft_prime_pow<2>(float2* v) { fft2(v[0], v[1]); }
template <> __device__ __forceinline__ void dft_prime_pow<3>(float2* v) { dft3(v[0], v[1], v[2]); }
template <> __device__ __forceinline__ void dft_prime_pow<4>(float2* v) { fft4(v[0], v[1], v[2], v[3]); }
template <> __device__ __forceinline__ void dft_prime_pow<5>(float2* v) { dft5(v[0], v[1], v[2], v[3], v[4]); }
template <> __device__ __forceinline__ void dft_prime_pow<8>(float2* v) { float2(&a)[8] = *reinterpret_cast<float2(*)[8]>(v); fft8(a); }
template <> __device__ __forceinline__ void dft_prime_pow<9>(float2* v) { // 3 x 3 Cooley-Tukey: n = 3 n1 + n2, k = k1 + 3 k2
    float2 a[3][3];
#pragma unroll
    for (int n2 = 0; n2 < 3; ++n2) {
        a[n2][0] = v[n2]; a[n2][1] = v[3 + n2]; a[n2][2] = v[6 + n2];
        dft3(a[n2][0], a[n2][1], a[n2][2]); // over n1 -> k1
    }
    constexpr float c1 = 0.76604444311897803520f, s1 = 0.64278760968653932632f;  // 2 pi / 9
    constexpr float c2 = 0.17364817766693034885f, s2 = 0.98480775301220805937f;  // 4 pi / 9
    constexpr float c4 = -0.93969262078590838405f, s4 = 0.34202014332566873304f; // 8 pi / 9
    a[1][1] = cmulf(a[1][1], make_float2(c1, -s1)); // W_9^{n2 k1}
    a[1][2] = cmulf(a[1][2], make_float2(c2, -s2));
    a[2][1] = cmulf(a[2][1], make_float2(c2, -s2));
    a[2][2] = cmulf(a[2][2], make_float2(c4, -s4));
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
        dft3(a[0][k1], a[1][k1], a[2][k1]); // over n2 -> k2
        v[k1] = a[0][k1]; v[k1 + 3] = a[1][k1]; v[k1 + 6] = a[2][k1];
    }
}
template <> __device__ __forceinline__ void dft_prime_pow<16>(float2* v) {
    fft16<1>(v);
    float2 t[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) t[q] = v[perm16(q)];
#pragma unroll
    for (int q = 0; q < 16; ++q) v[q] = t[q];
}
// prime-factor (Good-Thomas) DFT of R = N1 N2, gcd(N1, N2) = 1: input n = (N2 n1 + N1 n2) mod R, output k with k = k1 (mod N1), k = k2 (mod N2); no twiddles
template <int N1, int N2>
__device__ __forceinline__ void dft_pfa(float2* v) {
    constexpr int R = N1 * N2;
    float2        a[N2][N1];
#pragma unroll
    for (int n2 = 0; n2 < N2; ++n2) {
#pragma unroll
        for (int n1 = 0; n1 < N1; ++n1) a[n2][n1] = v[(N2 * n1 + N1 * n2) % R];
        dft_prime_pow<N1>(a[n2]); // over n1 -> k1
    }
#pragma unroll
    for (int k1 = 0; k1 < N1; ++k1) {
        float2 b[N2];
#pragma unroll
        for (int n2 = 0; n2 < N2; ++n2) b[n2] = a[n2][k1];
        dft_prime_pow<N2>(b); // over n2 -> k2
#pragma unroll
        for (int k2 = 0; k2 < N2; ++k2) {
            int k = 0; // the k below R with k = k1 (mod N1) and k = k2 (mod N2): found at compile time after unrolling
#pragma unroll
            for (int c = 0; c < R; ++c)
                if (c % N1 == k1 && c % N2 == k2) k = c;
            v[k] = b[k2];
        }
    }
}
template <int R>
__device__ __forceinline__ void dft_any(float2* v) {
    if constexpr (R == 6) dft_pfa<2, 3>(v);
    else if constexpr (R == 10) dft_pfa<2, 5>(v);
    else if constexpr (R == 12) dft_pfa<4, 3>(v);
    else if constexpr (R == 15) dft_pfa<3, 5>(v);
    else dft_prime_pow<R>(v);
}

// one autosort pass of radix R on a frame in LDS (in place; two barriers), butterflies t, t + tf, ... of the frame's tf lanes
// (not inlined: eleven radices in one kernel body cost 28 spilled VGPRs and 114 spilled SGPRs; a call per pass and frame group is nothing)
template <int R>
__device__ __attribute__((noinline)) void smooth_pass(float2* buf, int t, int tf, int N, int p, const float2* tw /*the W_N^j table, in LDS*/) {
    auto BARRIER = [] { __syncthreads(); };
    constexpr int NBL = 16 / R; // butterflies a lane can hold
    const int     NB = N / R, su = N / (p * R);
    float2        v[NBL][R];
    int           kk[NBL];
#pragma unroll
    for (int b = 0; b < NBL; ++b) {
        const int i = t + b * tf;
        kk[b] = 0;
        if (i < NB) {
            const int k = p > 1 ? i % p : 0;
            kk[b] = k;
#pragma unroll
            for (int r = 0; r < R; ++r) v[b][r] = buf[i + r * NB];
            if (k > 0) { // W_{pR}^{r k} = tw[r k su]: exact W^k and W^{2k}, the higher powers from two interleaved chains
                const float2 w1 = tw[k * su];
                v[b][1] = cmulf(v[b][1], w1);
                if constexpr (R > 2) {
                    const float2 w2 = tw[2 * k * su];
                    float2       wo = w1, we = w2;
                    v[b][2] = cmulf(v[b][2], we);
#pragma unroll
                    for (int r = 3; r < R; r += 2) {
                        wo      = cmulf(wo, w2);
                        v[b][r] = cmulf(v[b][r], wo);
                        if (r + 1 < R) {
                            we          = cmulf(we, w2);
                            v[b][r + 1] = cmulf(v[b][r + 1], we);
                        }
                    }
                }
            }
        }
    }
    BARRIER();
#pragma unroll
    for (int b = 0; b < NBL; ++b) {
        const int i = t + b * tf;
        if (i < NB) {
            dft_any<R>(v[b]);
            const int base = (i - kk[b]) * R + kk[b];
#pragma unroll
            for (int r = 0; r < R; ++r) buf[base + r * p] = v[b][r];
        }
    }
    BARRIER();
}

// fpb frames per workgroup iteration, tf lanes per frame (plan), persistent
template <int BS>
__global__ __launch_bounds__(BS, 4) void fft_smooth_kernel(const float* __restrict__ in, const float* __restrict__ window, const float2* __restrict__ tw, FftPlanDev plan, FftOutputs out,
                                                          long n_frames) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int  N = plan.N, tf = plan.tf;
    const int  fl = threadIdx.x / tf, t = fl < plan.fpb ? threadIdx.x - fl * tf : N; // (lanes beyond the last frame slot own no butterfly: every index test fails for t = N)
    float2*    twl = lds;                                                          // the twiddle table W_N^j, once per (persistent) workgroup
    float2*    buf = lds + N + (size_t)(fl < plan.fpb ? fl : 0) * N;
    for (int i = threadIdx.x; i < N; i += blockDim.x) twl[i] = tw[i];
    const long ngroups = (n_frames + plan.fpb - 1) / plan.fpb;
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        const long frame = g * plan.fpb + fl;
        const bool live  = frame < n_frames && fl < plan.fpb;
        if (live) { // load + window (fft.hpp:148-162); real input becomes (x w, 0)
            if (out.real_input) {
                const float* x = in + frame * N;
                for (int i = t; i < N; i += tf) buf[i] = make_float2(x[i] * (window ? window[i] : 1.f), 0.f);
            } else {
                const float2* x = reinterpret_cast<const float2*>(in) + frame * N;
                for (int i = t; i < N; i += tf) {
                    float2 s = x[i];
                    if (window) { const float w = window[i]; s.x *= w; s.y *= w; }
                    buf[i] = s;
                }
            }
        }
        __syncthreads();
        int p = 1;
        for (int pass = 0; pass < plan.npass; ++pass) {
            const int R = plan.radix[pass];
            switch (R) { // wave-uniform
            case 2: smooth_pass<2>(buf, t, tf, N, p, twl); break;
            case 3: smooth_pass<3>(buf, t, tf, N, p, twl); break;
            case 4: smooth_pass<4>(buf, t, tf, N, p, twl); break;
            case 5: smooth_pass<5>(buf, t, tf, N, p, twl); break;
            case 6: smooth_pass<6>(buf, t, tf, N, p, twl); break;
            case 8: smooth_pass<8>(buf, t, tf, N, p, twl); break;
            case 9: smooth_pass<9>(buf, t, tf, N, p, twl); break;
            case 10: smooth_pass<10>(buf, t, tf, N, p, twl); break;
            case 12: smooth_pass<12>(buf, t, tf, N, p, twl); break;
            case 15: smooth_pass<15>(buf, t, tf, N, p, twl); break;
            default: smooth_pass<16>(buf, t, tf, N, p, twl); break;
            }
            p *= R;
        }
        if (live)
            for (int k = t; k < N; k += tf) emit_bin(out, frame, N, k, buf[k]);
        __syncthreads(); // every lane is done with the frame before the next one lands on it
    }
}

// ------------------------------------------------------------------------------------------------ compile-time plans for the common sizes
// The run-time kernel above pays for its generality: a call per pass, an integer division per butterfly, run-time strides.  The sizes an SDR chain actually
// asks for (1000, 1200, 1536, 1920, 2000, 3000, 3072, 4000, 5000, 6000, 6144, 8000, ...) get the same passes with everything known at compile time -- as the
// power-of-two kernels do: the first pass reads its inputs straight from global memory (window fused, coalesced), the last emits every output straight from
// registers, a frame makes npass - 1 LDS round trips, twiddles from the LDS table + two power chains.
#ifndef GR4_SMOOTH_GRID_MULT
#define GR4_SMOOTH_GRID_MULT 1 // persistent grid = resident workgroups x this
#endif
#ifndef GR4_SMOOTH_PTS
#define GR4_SMOOTH_PTS 16 // points a lane holds between the barriers of a pass (compile-time plans)
#endif
template <int N, int R0, int R1, int R2, int R3>
struct SmoothCT {
    static constexpr int kR[4] = {R0, R1, R2, R3};
    static constexpr int NP    = 1 + (R1 > 1) + (R2 > 1) + (R3 > 1);
    static constexpr int cdiv(int a, int b) { return (a + b - 1) / b; }
    static constexpr int tf_of(int R) { return R > 1 ? cdiv(N, R * (GR4_SMOOTH_PTS / R)) : 1; }
    static constexpr int cmax(int a, int b) { return a > b ? a : b; }
    static constexpr int TF  = cmax(cmax(tf_of(R0), tf_of(R1)), cmax(tf_of(R2), tf_of(R3)));
    static constexpr int FPB = cmax(1, (512 / TF) < (int)(40 * 1024 / (N * 8)) ? (512 / TF) : (int)(40 * 1024 / (N * 8)));
    static constexpr int BS  = cdiv(TF * FPB, 64) * 64;
    // twiddle table: a pass of radix R reads W_N^j for j = k SU and 2 k SU < 2 N / R only, and the first pass none: 2 N / (smallest later radix) entries instead
    // of N (at N = 6000 the difference between one and two workgroups per CU)
    static constexpr int need(int R) { return R > 2 ? (2 * N) / R : (R == 2 ? N / 2 : 0); } // (radix 2 multiplies by W^k only)
    static constexpr int TW  = cmax(cmax(need(R1), need(R2)), need(R3)) + 2;
    static constexpr size_t LDS = ((size_t)FPB * N + TW) * sizeof(float2);
    static_assert(R0 * R1 * R2 * R3 == N && BS <= 1024, "plan");
};

template <int N, int R, int P /*product of the radices before this pass*/, bool FIRST, bool LAST, int TF>
__device__ __forceinline__ void smooth_ct_pass(float2* buf, int t, const float2* twl, const float* __restrict__ x, const float* __restrict__ window, const FftOutputs& out, long frame, bool live) {
    constexpr int NBL = GR4_SMOOTH_PTS / R, NB = N / R, SU = N / (P * R);
    float2        v[NBL][R];
    int           kk[NBL];
#pragma unroll
    for (int b = 0; b < NBL; ++b) {
        const int i = t + b * TF;
        kk[b] = 0;
        if (i < NB) {
            if constexpr (FIRST) {
                if (live) {
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        const int   n = i + r * NB;
                        const float w = window ? window[n] : 1.f;
                        if (out.real_input) v[b][r] = make_float2(x[n] * w, 0.f);
                        else { const float2 s_ = reinterpret_cast<const float2*>(x)[n]; v[b][r] = make_float2(s_.x * w, s_.y * w); }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < R; ++r) v[b][r] = make_float2(0.f, 0.f);
                }
            } else {
                const int k = i % P;
                kk[b] = k;
#pragma unroll
                for (int r = 0; r < R; ++r) v[b][r] = buf[i + r * NB];
                if (k > 0) { // W_{PR}^{r k} = tw[r k SU]
                    const float2 w1 = twl[k * SU];
                    v[b][1] = cmulf(v[b][1], w1);
                    if constexpr (R > 2) {
                        const float2 w2 = twl[2 * k * SU];
                        float2       wo = w1, we = w2;
                        v[b][2] = cmulf(v[b][2], we);
#pragma unroll
                        for (int r = 3; r < R; r += 2) {
                            wo      = cmulf(wo, w2);
                            v[b][r] = cmulf(v[b][r], wo);
                            if (r + 1 < R) {
                                we          = cmulf(we, w2);
                                v[b][r + 1] = cmulf(v[b][r + 1], we);
                            }
                        }
                    }
                }
            }
        }
    }
    if constexpr (!FIRST && !LAST) __syncthreads(); // in place: everybody has read before anybody writes
#pragma unroll
    for (int b = 0; b < NBL; ++b) {
        const int i = t + b * TF;
        if (i < NB) {
            dft_any<R>(v[b]);
            if constexpr (LAST) {
                if (live) {
#pragma unroll
                    for (int r = 0; r < R; ++r) emit_bin(out, frame, N, i + r * NB, v[b][r]); // P R = N: k = i, the outputs are i + r N / R
                }
            } else {
                const int base = (i - kk[b]) * R + kk[b];
#pragma unroll
                for (int r = 0; r < R; ++r) buf[base + r * P] = v[b][r];
            }
        }
    }
    if constexpr (!LAST) __syncthreads();
}

template <int N, int R0, int R1, int R2, int R3>
__global__ __launch_bounds__((SmoothCT<N, R0, R1, R2, R3>::BS), 2) void fft_smooth_ct_kernel(const float* __restrict__ in, const float* __restrict__ window, const float2* __restrict__ tw, FftOutputs out,
                                                                                               long n_frames) {
    using PL = SmoothCT<N, R0, R1, R2, R3>;
    constexpr int TF = PL::TF, FPB = PL::FPB;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    float2* twl = lds;
    for (int i = threadIdx.x; i < PL::TW && i < N; i += blockDim.x) twl[i] = tw[i];
    __syncthreads();
    const long ngroups = (n_frames + FPB - 1) / FPB;
    for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
        int tid = threadIdx.x;
        asm volatile("" : "+v"(tid)); // (lane-dependent offsets recomputed per iteration instead of hoisted into dozens of registers)
        const int  fl = tid / TF, t = fl < FPB ? tid - fl * TF : N;
        float2*    buf = lds + PL::TW + (size_t)(fl < FPB ? fl : 0) * N;
        const long frame = g * FPB + fl;
        const bool live = frame < n_frames && fl < FPB;
        const float* xf = in + (live ? frame : 0) * N * (out.real_input ? 1 : 2);
        constexpr bool two = PL::NP == 2, three = PL::NP == 3, four = PL::NP == 4;
        static_assert(two || three || four, "two to four passes");
        smooth_ct_pass<N, R0, 1, true, false, TF>(buf, t, twl, xf, window, out, frame, live);
        if constexpr (two) {
            smooth_ct_pass<N, R1, R0, false, true, TF>(buf, t, twl, xf, window, out, frame, live);
        } else {
            smooth_ct_pass<N, R1, R0, false, false, TF>(buf, t, twl, xf, window, out, frame, live);
            if constexpr (three) {
                smooth_ct_pass<N, R2, R0 * R1, false, true, TF>(buf, t, twl, xf, window, out, frame, live);
            } else {
                smooth_ct_pass<N, R2, R0 * R1, false, false, TF>(buf, t, twl, xf, window, out, frame, live);
                smooth_ct_pass<N, R3, R0 * R1 * R2, false, true, TF>(buf, t, twl, xf, window, out, frame, live);
            }
        }
        __syncthreads(); // every lane is done with the frame before the next one lands on it
    }
}

// The sizes with a compile-time plan: fft_smooth_sizes.inc, X(N, R0, R1, R2, R3) with the radices in pass order (1 = no such pass) -- every {2,3,5}-smooth size
// 18 .. 8192 of two to four passes (144 of them).  Generated: tools/gen_smooth_plans.py proposes two factorisations per size (the lexicographically largest radix
// tuple / the largest smallest radix; first pass never a power of two where avoidable, largest radix second, smallest last), tools/smooth_sweep.sh measures every
// size under both on an MI355X, and the faster one is written out (profiles/r03_fft_smooth_plan_sweep.json has the table).  No rule predicts the winner: 3000 wants
// 10-15-10-2 (230 against 188 Gsamples/s for 15-8-5-5), 6000 wants 10-10-10-6 (228 against 184 for 15-16-5-5), 720 wants 10-12-6 (278 against 194 for 10-9-8).

template <int N, int R0, int R1, int R2, int R3>
inline int fft_smooth_ct_launch(const float* d_in, const float* d_window, const float2* d_tw, const FftOutputs& o, long n_frames, hipStream_t st) {
    using PL = SmoothCT<N, R0, R1, R2, R3>;
    auto kern = fft_smooth_ct_kernel<N, R0, R1, R2, R3>;
    static PerDevice per_device;
    bool             first = false;
    int              dev = -1, n_cu = per_device.current(&first, &dev);
    GR4_REQUIRE(n_cu != 0, "fft: cannot query the current device");
    if (first) {
        n_cu = -n_cu;
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)PL::LDS));
        per_device.done(dev, n_cu);
    }
    const long groups = ceil_div(n_frames, (long)PL::FPB);
    const long per_cu = std::max<long>(1, std::min<long>(2048 / PL::BS, (long)(160 * 1024 / PL::LDS)));
#ifdef GR4_SMOOTH_ONESHOT // developer experiment: one workgroup per frame group instead of a persistent grid
    (void)per_cu;
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min<long>(groups, 1L << 30)), dim3(PL::BS), PL::LDS, st, d_in, d_window, d_tw, o, n_frames);
#else
    hipLaunchKernelGGL(kern, dim3((unsigned)std::min<long>(groups, (long)n_cu * per_cu * GR4_SMOOTH_GRID_MULT)), dim3(PL::BS), PL::LDS, st, d_in, d_window, d_tw, o, n_frames);
#endif
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}
// GR4HIP_UNSUPPORTED: no compile-time plan for this size (the run-time kernel takes it)
inline int fft_smooth_ct_dispatch(int N, const float* d_in, const float* d_window, const float2* d_tw, const FftOutputs& o, long n_frames, hipStream_t st) {
    switch (N) {
#define X(N_, A, B, C, D) case N_: return fft_smooth_ct_launch<N_, A, B, C, D>(d_in, d_window, d_tw, o, n_frames, st);
#include "fft_smooth_sizes.inc"
#undef X
    default: return GR4HIP_UNSUPPORTED;
    }
}

// {2,3,5}-smooth and not a power of two?
inline bool fft_is_smooth235(size_t N) {
    if (N < 2) return false;
    size_t m = N;
    for (size_t f : {2, 3, 5})
        while (m % f == 0) m /= f;
    return m == 1;
}
// the fewest passes with radices from {16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2}, large radices first (the first pass has no twiddles)
inline int fft_build_smooth_plan(size_t N, FftPlanDev* plan) {
    static const int kR[] = {16, 15, 12, 10, 9, 8, 6, 5, 4, 3, 2};
    std::vector<int> best((size_t)N + 1, 1 << 20), pick((size_t)N + 1, 0);
    best[1] = 0;
    for (size_t n = 2; n <= N; ++n) {
        if (N % n) continue;
        for (int r : kR)
            if (n % (size_t)r == 0 && best[n / (size_t)r] + 1 < best[n]) { best[n] = best[n / (size_t)r] + 1; pick[n] = r; }
    }
    if (best[N] > 15) return GR4HIP_UNSUPPORTED;
    plan->N = (int)N;
    int np = 0;
    for (size_t n = N; n > 1; n /= (size_t)pick[n]) plan->radix[np++] = pick[n];
    std::sort(plan->radix, plan->radix + np, [](int a, int b) { return a > b; });
    plan->npass = np;
    int tf = 1;
    for (int i = 0; i < np; ++i) {
        const int R = plan->radix[i], per = R * (16 / R); // points a lane covers in this pass
        tf = std::max(tf, (int)((N + (size_t)per - 1) / (size_t)per));
    }
    plan->tf  = tf;
    plan->fpb = std::max(1, std::min(512 / tf, (int)(40 * 1024 / (N * sizeof(float2))))); // <= 512 lanes and <= ~48 KB of LDS (frames + the twiddle table) per workgroup: several workgroups per CU
    return GR4HIP_OK;
}

inline int fft_smooth_launch(const FftPlanDev& plan, const float* d_in, const float* d_window, const float2* d_tw, const FftOutputs& o, long n_frames, hipStream_t st) {
    if (!dev_switch(kDevFftSmoothRuntime)) { // (developer switch: the run-time plan kernel for every size; the tests compare the two)
        const int rc = fft_smooth_ct_dispatch(plan.N, d_in, d_window, d_tw, o, n_frames, st);
        if (rc != GR4HIP_UNSUPPORTED) return rc;
    }
    const size_t lds = (size_t)(plan.fpb + 1) * plan.N * sizeof(float2); // frames + the twiddle table
    const int    bs  = ((plan.tf * plan.fpb + 63) / 64) * 64;
    static PerDevice per_device;
    bool             first = false;
    int              dev = -1, n_cu = per_device.current(&first, &dev);
    GR4_REQUIRE(n_cu != 0, "fft: cannot query the current device");
    if (first) {
        n_cu = -n_cu;
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fft_smooth_kernel<512>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fft_smooth_kernel<1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024));
        per_device.done(dev, n_cu);
    }
    const long groups = ceil_div(n_frames, (long)plan.fpb);
    const long per_cu = std::max<long>(1, std::min<long>(2048 / bs, (long)(160 * 1024 / std::max<size_t>(lds, 1))));
    const dim3 grid((unsigned)std::min<long>(groups, (long)n_cu * per_cu));
    if (bs <= 512) hipLaunchKernelGGL(fft_smooth_kernel<512>, grid, dim3(bs), lds, st, d_in, d_window, d_tw, plan, o, n_frames);
    else hipLaunchKernelGGL(fft_smooth_kernel<1024>, grid, dim3(bs), lds, st, d_in, d_window, d_tw, plan, o, n_frames);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

} // namespace gr4
