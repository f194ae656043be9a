// fft.hip -- the streaming FFT block on gfx950: C-ABI (gr4hip_fft_*), the generic LDS-resident Stockham kernel (radix 8/4/2, any power of
// two <= 8192), the multi-kernel paths (four-step for powers of two 16384..2^20, Bluestein for every other size <= 2^19), phase unwrap and DataSet ranges.
// The compile-time-plan kernels for N = 256..8192 (the fast path) are in fft_kernels.hpp.
//
// Replaces gr::blocks::fft::FFT<T>::processBulk (blocks/fourier/.../fft.hpp:147-171): window -> forward DFT
// (algorithm/.../fourier/fft.hpp:113-153) -> magnitude / phase / Re / Im (fft_common.hpp:20-123), many frames per
// launch instead of one frame per work() call.  One workgroup owns whole frames: the frame is read from HBM once
// (window fused into the load), all passes run in LDS, and every requested output is written once, coalesced.
// The fused FIR->FFT->mag2 kernels live in chain_fused.hip.
// (round 5) the transforms' frames come in once and their results leave once: streaming (nt) hints on both (buffer_ops.hpp).  A/B on one box, alternated, 2^27 points per launch
// (profiles/r05_streaming_hints.txt): N = 512 / 1024 mag2 472 - 479 -> 498 - 510 Gsamples/s (+ 4 ... + 7 %), spectrum 360 / 355 -> 367 - 374 / 355 - 375.
#ifndef GR4_BUF_STORE_AUX
#define GR4_BUF_STORE_AUX 2
#endif
#ifndef GR4_BUF_LOAD_AUX
#define GR4_BUF_LOAD_AUX 2
#endif
#include "fft_kernels.hpp"
#include "fft_smooth.hpp"

namespace gr4 {

template <int R>
__device__ __forceinline__ void load_bfly(float2* v, const float2* __restrict__ src, int i, int NB, int p, int step_unit, const float2* __restrict__ tw) {
    const int k = i & (p - 1);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float2 x = src[i + r * NB];
        if (r > 0 && p > 1) x = cmulf(x, tw[r * k * step_unit]);
        v[r] = x;
    }
}

template <int R>
__device__ __forceinline__ void store_bfly(const float2* v, float2* dst, int i, int p) {
    const int k    = i & (p - 1);
    const int base = (i - k) * R + k;
#pragma unroll
    for (int r = 0; r < R; ++r) dst[base + r * p] = v[r];
}

// the Stockham passes of one frame that sits in LDS (natural order in, natural order out); every lane of the workgroup calls it (it synchronises)
__device__ __forceinline__ void lds_passes(float2* buf, int t, int tf, const FftPlanDev& plan, const float2* __restrict__ tw) {
    const int N = plan.N;
    int p = 1;
    for (int pass = 0; pass < plan.npass; ++pass) {
        const int R  = plan.radix[pass];
        const int NB = N / R;
        const int nb = NB / tf > 0 ? NB / tf : 1;
        const int su = N / (p * R); // twiddle index unit: W_{pR}^{rk} = tw[r*k*su]
        float2    v[8];
        if (R == 8) {
            if (t < NB) load_bfly<8>(v, buf, t, NB, p, su, tw);
        } else if (R == 4) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (b < nb && t + b * tf < NB) load_bfly<4>(v + 4 * b, buf, t + b * tf, NB, p, su, tw);
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (b < nb && t + b * tf < NB) load_bfly<2>(v + 2 * b, buf, t + b * tf, NB, p, su, tw);
        }
        __syncthreads();
        if (R == 8) {
            if (t < NB) { fft8(v); store_bfly<8>(v, buf, t, p); }
        } else if (R == 4) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (b < nb && t + b * tf < NB) { fft4(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]); store_bfly<4>(v + 4 * b, buf, t + b * tf, p); }
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (b < nb && t + b * tf < NB) { fft2(v[2 * b], v[2 * b + 1]); store_bfly<2>(v + 2 * b, buf, t + b * tf, p); }
        }
        __syncthreads();
        p *= R;
    }
}

// One workgroup = fpb frames; tf = max(1, N/8) lanes per frame, 8 points per lane.
__global__ void fft_block_kernel(const float* __restrict__ in, const float* __restrict__ window, const float2* __restrict__ tw, FftPlanDev plan, FftOutputs out,
                                 long n_frames) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int  N     = plan.N;
    const int  tf    = plan.tf;
    const int  fl    = threadIdx.x / tf; // frame slot in block
    const int  t     = threadIdx.x - fl * tf;
    const long frame = (long)blockIdx.x * plan.fpb + fl;
    const bool live  = frame < n_frames;
    float2*    buf   = lds + (size_t)fl * N;

    // ---- load + window (fft.hpp:148-162); real input becomes (x*w, 0)
    if (live) {
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

    lds_passes(buf, t, tf, plan, tw);
    if (!live) return;

    // ---- epilogue over natural-order bins
    for (int k = t; k < N; k += tf) emit_bin(out, frame, N, k, buf[k]);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Sizes beyond one workgroup's LDS and sizes that are not a power of two run as short pipelines of the kernels above through HBM
// scratch buffers (3 - 5x the traffic of the single-kernel path; these are the rarely used corners of the block, not its hot path).
//
// (1) N = N1 * 4096 with N1 in {4, 8, 16} (16384, 32768, 65536), the four-step decomposition n = 4096 n1 + n2, k = k1 + N1 k2:
//       X[k1 + N1 k2] = sum_{n2} W_4096^{n2 k2} ( W_N^{n2 k1} sum_{n1} x[4096 n1 + n2] W_N1^{n1 k1} )
//     fft_big_cols_kernel: window, N1-point DFT down the columns in registers, twiddle, A[f][k1][n2]  (lanes over n2: coalesced)
//     fft_fast_kernel<12>: the N1 rows of every frame as 4096-point frames -> B[f][k1][k2]
//     fft_emit_kernel:     bin k = k1 + N1 k2 from B, every requested output (LDS-free: consecutive lanes take consecutive k, the
//                          N1-strided read of B stays inside 64-byte segments)
template <int N1>
__global__ __launch_bounds__(256) void fft_big_cols_kernel(const float* __restrict__ in, const float* __restrict__ window, const float2* __restrict__ twN /*W_N^j*/,
                                                            float2* __restrict__ A, long n_frames, int real_input) {
    constexpr int N2 = 4096, N = N1 * N2;
    const long    gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long    f   = gid / N2;
    const int     n2  = (int)(gid % N2);
    if (f >= n_frames) return;
    float2 v[N1];
#pragma unroll
    for (int n1 = 0; n1 < N1; ++n1) {
        const long i = f * N + (long)N2 * n1 + n2;
        float2     x = real_input ? make_float2(in[i], 0.f) : reinterpret_cast<const float2*>(in)[i];
        if (window) { const float w = window[N2 * n1 + n2]; x.x *= w; x.y *= w; }
        v[n1] = x;
    }
    if constexpr (N1 == 16) fft16<1>(v);
    else dft_small<N1>(v);
#pragma unroll
    for (int k1 = 0; k1 < N1; ++k1) {
        float2 y = v[N1 == 16 ? perm16(k1) : k1];
        if (k1 > 0) y = cmul(y, twN[(n2 * k1) & (N - 1)]); // n2 k1 < N
        A[(f * N1 + k1) * N2 + n2] = y;
    }
}

// every requested output from a spectrum buffer: bin k of frame f sits at B[f * N + (k % n1) * (N / n1) + k / n1] (n1 = 1: natural order);
// frame0: index of the buffer's first frame in the outputs (long inputs run in batches of frames through bounded scratch buffers)
__global__ __launch_bounds__(256) void fft_emit_kernel(const float2* __restrict__ B, FftOutputs out, int N, int n1, long n_frames, long frame0) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long f   = gid / N;
    const int  k   = (int)(gid % N);
    if (f >= n_frames) return;
    emit_bin(out, frame0 + f, N, k, B[f * N + (long)(k % n1) * (N / n1) + k / n1]);
}

// (1b, round 5) N = 65536 = 256 x 256 in TWO kernels, 32 B of HBM traffic per point instead of the 48 of the three-kernel pipeline above (bm_fft.cpp:44-58 times this size):
//       n = 256 n1 + n2, k = k1 + 256 k2:   X[k1 + 256 k2] = sum_{n2} W_256^{n2 k2} ( W_N^{n2 k1} sum_{n1} x[256 n1 + n2] W_256^{n1 k1} )
//     fft_64k_cols_kernel: a workgroup takes 16 neighbouring columns n2 of a frame (every row of them is one 128-byte segment), window, the 256-point transform down each
//                          column (16 lanes x 16 points, radix 16 x 16, one exchange through LDS), x W_N^{n2 k1}, A[f][k1][n2] (128-byte segments again)
//     fft_64k_rows_kernel: a workgroup takes 16 neighbouring rows k1 of A (2 KiB each), the 256-point transform along each, a transposition through LDS so that 16
//                          consecutive lanes hold 16 consecutive bins k1 + 256 k2, every requested output
// W_256^j is read as W_N^{256 j} from the one twiddle table of the size.
__device__ __forceinline__ void fft256_by_16_lanes(float2 (&v)[16], float2* S /*this transform's 16 x 17 exchange rows*/, int t, const float2* w256 /*W_256^j, in LDS*/) {
    fft16<1>(v); // over r: Y_t[q] at slot perm16(q)
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        float2 y = v[perm16(q)];
        if (q > 0) y = cmul(y, w256[(t * q) & 255]); // W_256^{t q}
        S[q * 17 + t] = y;
    }
    __syncthreads();
#pragma unroll
    for (int tt = 0; tt < 16; ++tt) v[tt] = S[t * 17 + tt]; // lane q = t gathers Y_tt[q]
    fft16<1>(v); // over t: X[q + 16 p] at slot perm16(p)
}
__global__ __launch_bounds__(256) void fft_64k_cols_kernel(const float* __restrict__ in, const float* __restrict__ window, const float2* __restrict__ twG /*W_N^{n2 k1} at [k1][n2]*/,
                                                            const float2* __restrict__ tw4096, float2* __restrict__ A, int real_input) {
    __shared__ float2 S[16 * 16 * 17];
    __shared__ float2 w256[256];
    w256[threadIdx.x] = tw4096[16 * threadIdx.x]; // W_256^j = W_4096^{16 j}
    constexpr int N = 65536;
    const long    f = blockIdx.x >> 4;
    const int     cb = blockIdx.x & 15, col = threadIdx.x & 15, t = threadIdx.x >> 4, n2 = 16 * cb + col;
    float2        v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int  n1 = t + 16 * r;
        const long i  = f * N + 256L * n1 + n2;
        float2     x  = real_input ? make_float2(in[i], 0.f) : reinterpret_cast<const float2*>(in)[i];
        if (window) { const float w = window[256 * n1 + n2]; x.x *= w; x.y *= w; }
        v[r] = x;
    }
    __syncthreads();
    fft256_by_16_lanes(v, S + col * 16 * 17, t, w256);
    // W_N^{n2 k1} from a table laid out [k1][n2]: the 16 lanes of a column group read 128 consecutive bytes, like the samples.  (Measured: the same values gathered from the
    // linear table W_N^j -- one random 8-byte read per point -- 95 Gsamples/s; as W_N^{n2 t} x W_4096^{n2 p}, two rounded factors -- 150, but 1.2e-5 on the strong-tone test
    // signal where one exact factor gives 8e-6.)
#pragma unroll
    for (int p = 0; p < 16; ++p) {
        const int k1 = t + 16 * p;
        A[(f * 256 + k1) * 256 + n2] = cmul(v[perm16(p)], twG[k1 * 256 + n2]);
    }
}
__global__ __launch_bounds__(256) void fft_64k_rows_kernel(const float2* __restrict__ A, const float2* __restrict__ tw4096, FftOutputs out, long frame0) {
    __shared__ float2 S[16 * 16 * 17];
    __shared__ float2 w256[256];
    w256[threadIdx.x] = tw4096[16 * threadIdx.x];
    constexpr int N = 65536;
    const long    f = blockIdx.x >> 4;
    const int     rb = blockIdx.x & 15, row = threadIdx.x >> 4, t = threadIdx.x & 15;
    const float2* a = A + (f * 256 + 16 * rb + row) * 256;
    float2        v[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = a[t + 16 * r];
    __syncthreads();
    fft256_by_16_lanes(v, S + row * 16 * 17, t, w256);
    __syncthreads(); // every transform has read its exchange rows: S is free for the transposition
    // X[k2 = t + 16 p] of row k1 = 16 rb + row -> O[k2][row]; then lane l emits bins (16 rb + (l & 15)) + 256 k2, k2 = (l >> 4) + 16 i: 16 consecutive lanes = 16 consecutive bins
#pragma unroll
    for (int p = 0; p < 16; ++p) S[(t + 16 * p) * 17 + row] = v[perm16(p)];
    __syncthreads();
    const int rr = threadIdx.x & 15, k2b = threadIdx.x >> 4;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int k2 = k2b + 16 * i;
        emit_bin(out, frame0 + f, N, 16 * rb + rr + 256 * k2, S[k2 * 17 + rr]);
    }
}

// N1 = 32 .. 256 (N = 2^17 .. 2^20): the column transforms no longer fit a lane's registers.  They run as ordinary N1-point frames of the block
// kernels, between two transpositions done in 32 x 32 tiles through LDS (reads and writes both in 256-byte rows):
//   gather:  x[f][n1][n2] (window applied) -> Xt[f][n2][n1]        cols: N1-point transforms of the 4096 rows of Xt -> Yt[f][n2][k1]
//   scatter: Yt[f][n2][k1] W_N^{n2 k1} -> A[f][k1][n2]              rows: as above                    emit: B[f][k1][k2] -> bin k1 + N1 k2, tiled
struct BigTile { // this block's tile: rows a0 .. a0 + 31 of the N1 axis, columns b0 .. b0 + 31 of the 4096 axis, of frame f
    long f;
    int  a0, b0, tx, ty;
    __device__ BigTile(int n1) {
        const long per = 128L * (n1 / 32);
        f              = blockIdx.x / per;
        const int tl   = (int)(blockIdx.x % per);
        a0             = (tl / 128) * 32;
        b0             = (tl % 128) * 32;
        tx             = threadIdx.x & 31;
        ty             = threadIdx.x >> 5;
    }
};
__global__ __launch_bounds__(256) void fft_big_gather_kernel(const float* __restrict__ in, const float* __restrict__ window, float2* __restrict__ Xt, int n1, int real_input) {
    __shared__ float2 tile[32][33];
    const BigTile     t(n1);
    const long        N = 4096L * n1;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int  a = t.a0 + t.ty + 8 * r, b = t.b0 + t.tx;
        const long i = t.f * N + 4096L * a + b;
        float2     x = real_input ? make_float2(in[i], 0.f) : reinterpret_cast<const float2*>(in)[i];
        if (window) { const float w = window[4096L * a + b]; x.x *= w; x.y *= w; }
        tile[t.ty + 8 * r][t.tx] = x;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) Xt[(t.f * 4096 + t.b0 + t.ty + 8 * r) * n1 + t.a0 + t.tx] = tile[t.tx][t.ty + 8 * r];
}
__global__ __launch_bounds__(256) void fft_big_scatter_kernel(const float2* __restrict__ Yt, const float2* __restrict__ twN, float2* __restrict__ A, int n1) {
    __shared__ float2 tile[32][33];
    const BigTile     t(n1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int b = t.b0 + t.ty + 8 * r, k = t.a0 + t.tx; // b k < 4096 N1 = N
        tile[t.ty + 8 * r][t.tx] = cmul(Yt[(t.f * 4096 + b) * n1 + k], twN[(long)b * k]);
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) A[(t.f * n1 + t.a0 + t.ty + 8 * r) * 4096 + t.b0 + t.tx] = tile[t.tx][t.ty + 8 * r];
}
__global__ __launch_bounds__(256) void fft_big_emit_kernel(const float2* __restrict__ B, FftOutputs out, int n1, long frame0) {
    __shared__ float2 tile[32][33];
    const BigTile     t(n1);
#pragma unroll
    for (int r = 0; r < 4; ++r) tile[t.ty + 8 * r][t.tx] = B[(t.f * n1 + t.a0 + t.ty + 8 * r) * 4096 + t.b0 + t.tx];
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 4; ++r) emit_bin(out, frame0 + t.f, 4096 * n1, (t.a0 + t.tx) + n1 * (t.b0 + t.ty + 8 * r), tile[t.tx][t.ty + 8 * r]);
}

// (2) any other N <= 2^19 (the reference's Bluestein branch, algorithm/.../fourier/fft.hpp:353-381): X[k] = c*[k] sum_n (x[n] w[n] c*[n]) c[k-n],
//     c[n] = e^{+i pi n^2 / N}; the convolution is circular of length M = bit_ceil(2N - 1) <= 2^20 and runs through the power-of-two paths:
//     bluestein_pre -> FFT_M -> x FFT_M(c) and conjugate -> FFT_M (inverse through conjugation) -> bluestein_post (conj, 1/M, c*[k]) + outputs
__global__ __launch_bounds__(256) void bluestein_pre_kernel(const float* __restrict__ in, const float* __restrict__ window, const float2* __restrict__ cconj,
                                                             float2* __restrict__ a, int N, int M, long n_frames, int real_input) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long f   = gid / M;
    const int  n   = (int)(gid % M);
    if (f >= n_frames) return;
    float2 v = make_float2(0.f, 0.f);
    if (n < N) {
        const long i = f * (long)N + n;
        v            = real_input ? make_float2(in[i], 0.f) : reinterpret_cast<const float2*>(in)[i];
        if (window) { const float w = window[n]; v.x *= w; v.y *= w; }
        v = cmul(v, cconj[n]);
    }
    a[f * (long)M + n] = v;
}
__global__ __launch_bounds__(256) void bluestein_mul_kernel(float2* __restrict__ a, const float2* __restrict__ Bf, int M, long total) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const float2 p = cmul(a[gid], Bf[gid % M]);
    a[gid]         = make_float2(p.x, -p.y); // conjugate: the next forward transform is the inverse one
}
__global__ __launch_bounds__(256) void bluestein_post_kernel(const float2* __restrict__ a, const float2* __restrict__ cconj, FftOutputs out, int N, int M, long n_frames, long frame0) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    const long f   = gid / N;
    const int  k   = (int)(gid % N);
    if (f >= n_frames) return;
    const float2 r = a[f * (long)M + k];
    const float  s = 1.f / (float)M;
    emit_bin(out, frame0 + f, N, k, cmul(make_float2(r.x * s, -r.y * s), cconj[k]));
}

// N <= 4096 (M <= 8192): the whole chirp convolution of a frame in LDS -- chirp x window x input, FFT_M, x FFT_M(c) and conjugate, FFT_M again (the inverse through
// conjugation), conj / M x chirp, outputs.  12 .. 24 B of HBM traffic per sample instead of the ~160 B of the five-kernel pipeline above.
__global__ void bluestein_fused_kernel(const float* __restrict__ in, const float* __restrict__ window, const float2* __restrict__ cconj, const float2* __restrict__ Bf,
                                       const float2* __restrict__ tw /*W_M^j*/, FftPlanDev plan /*of M*/, FftOutputs out, int N, long n_frames) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int  M = plan.N, tf = plan.tf;
    const int  fl = threadIdx.x / tf, t = threadIdx.x - fl * tf;
    const long frame = (long)blockIdx.x * plan.fpb + fl;
    const bool live  = frame < n_frames;
    float2*    buf   = lds + (size_t)fl * M;
    for (int i = t; i < M; i += tf) {
        float2 v = make_float2(0.f, 0.f);
        if (live && i < N) {
            const long g = frame * (long)N + i;
            v            = out.real_input ? make_float2(in[g], 0.f) : reinterpret_cast<const float2*>(in)[g];
            if (window) { const float w = window[i]; v.x *= w; v.y *= w; }
            v = cmulf(v, cconj[i]);
        }
        buf[i] = v;
    }
    __syncthreads();
    lds_passes(buf, t, tf, plan, tw);
    for (int i = t; i < M; i += tf) {
        const float2 p = cmulf(buf[i], Bf[i]);
        buf[i]         = make_float2(p.x, -p.y);
    }
    __syncthreads();
    lds_passes(buf, t, tf, plan, tw);
    if (!live) return;
    const float sc = 1.f / (float)M;
    for (int k = t; k < N; k += tf) {
        const float2 r = buf[k];
        emit_bin(out, frame, N, k, cmulf(make_float2(r.x * sc, -r.y * sc), cconj[k]));
    }
}

// (round 5) the same chirp convolution on the compile-time 16 x 16 x R3 plan of the power-of-two kernels (fft_small_passes, fft_radix.hpp): 16 points per lane, M / 16 lanes
// per frame, 512 / (M / 16) frames per workgroup.  The two M-point transforms run register to register -- lane t's outputs of the first, bins t + j M / 16, ARE its inputs of
// the second -- so the frame only crosses LDS inside the transforms' own exchanges (the generic radix-8 passes above: four exchanges per transform at M = 2048, each through
// a full round trip).  bm_fft.cpp:44 times N = 1009: 58 -> see profiles/r05_bm_fft.txt.
template <int LOG2M>
__global__ __launch_bounds__(512) void bluestein_fast_kernel(const float* __restrict__ in, const float* __restrict__ window, const float2* __restrict__ cconj, const float2* __restrict__ Bf,
                                                             const float2* __restrict__ tw /*W_M^j*/, FftOutputs out, int N, long n_frames) {
    constexpr int M = 1 << LOG2M, TF = M / 16, FPB = 512 / TF, NPF = M + M / 32;
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int  fl = threadIdx.x / TF, tt = threadIdx.x % TF;
    const long frame = (long)blockIdx.x * FPB + fl;
    const bool live  = frame < n_frames;
    float2*    fb    = lds + (size_t)fl * NPF;
    const float2 sw2a = tw[(tt & 15) * (M / 256)], sw2b = tw[2 * (tt & 15) * (M / 256)], sw3 = tw[tt & 255], sw3sq = tw[(2 * (tt & 255)) & (M - 1)];
    float2 v[16], X[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int i = tt + r * TF;
        float2    x = make_float2(0.f, 0.f);
        if (live && i < N) {
            const long g = frame * (long)N + i;
            x            = out.real_input ? make_float2(in[g], 0.f) : reinterpret_cast<const float2*>(in)[g];
            if (window) { const float w = window[i]; x.x *= w; x.y *= w; }
            x = cmulf(x, cconj[i]);
        }
        v[r] = x;
    }
    fft_small_passes<LOG2M>(v, fb, tt, sw2a, sw2b, sw3, sw3sq, X, [] { __syncthreads(); });
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const float2 p = cmulf(X[j], Bf[tt + j * TF]);
        v[j]           = make_float2(p.x, -p.y);
    }
    fft_small_passes<LOG2M>(v, fb, tt, sw2a, sw2b, sw3, sw3sq, X, [] { __syncthreads(); });
    if (!live) return;
    const float sc = 1.f / (float)M;
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        const int k = tt + j * TF;
        if (k < N) emit_bin(out, frame, N, k, cmulf(make_float2(X[j].x * sc, -X[j].y * sc), cconj[k]));
    }
}
template <int LOG2M>
static int bluestein_fast_launch(const float* d_in, const float* win, const float2* cconj, const float2* Bf, const float2* tw, const FftOutputs& o, int N, long n_frames, hipStream_t st) {
    constexpr int    M = 1 << LOG2M, FPB = 512 / (M / 16);
    constexpr size_t lds = (size_t)FPB * (M + M / 32) * sizeof(float2);
    static PerDevice per_device; //  the LDS opt-in once per device and instantiation, not a driver call on every launch
    bool             first = false;
    int              dev = -1, n_cu = per_device.current(&first, &dev);
    GR4_REQUIRE(n_cu != 0, "fft: cannot query the current device");
    if (first) {
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(bluestein_fast_kernel<LOG2M>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        per_device.done(dev, -n_cu);
    }
    hipLaunchKernelGGL(bluestein_fast_kernel<LOG2M>, dim3((unsigned)ceil_div(n_frames, (long)FPB)), dim3(512), lds, st, d_in, win, cconj, Bf, tw, o, N, n_frames);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

// fft_common.hpp:71-89 unwrapPhase + :113-120 (deg, shift).  One workgroup per frame; wrap counts are integers, so a
// parallel prefix sum of the per-bin jump decisions reproduces the sequential loop.
__global__ void unwrap_kernel(const float* __restrict__ raw, float* __restrict__ out, int nout, int in_deg, int shift) {
    extern __shared__ int sc[]; // [blockDim.x]
    const float  pi    = 3.14159265358979323846f;
    const long   frame = blockIdx.x;
    const float* r     = raw + frame * nout;
    float*       o     = out + frame * nout;
    const int    per   = (nout + blockDim.x - 1) / blockDim.x;
    const int    k0    = threadIdx.x * per;
    int          local = 0;
    for (int k = k0; k < k0 + per && k < nout; ++k) {
        if (k == 0) continue;
        const float d = r[k] - r[k - 1];
        local += (d > pi) ? -1 : (d < -pi) ? 1 : 0;
    }
    sc[threadIdx.x] = local;
    __syncthreads();
    for (int off = 1; off < (int)blockDim.x; off <<= 1) { // inclusive Hillis-Steele scan
        const int v = (int)threadIdx.x >= off ? sc[threadIdx.x - off] : 0;
        __syncthreads();
        sc[threadIdx.x] += v;
        __syncthreads();
    }
    int c = sc[threadIdx.x] - local; // exclusive prefix for this lane's first bin
    for (int k = k0; k < k0 + per && k < nout; ++k) {
        if (k > 0) {
            const float d = r[k] - r[k - 1];
            c += (d > pi) ? -1 : (d < -pi) ? 1 : 0;
        }
        float ph = (float)((double)r[k] + (double)c * (double)(2.f * pi));
        if (in_deg) ph = ph * 180.f * 0.318309886183790671538f;
        const int ko = shift ? ((k + nout - nout / 2) % nout) : k; // std::rotate by nout/2
        o[ko]        = ph;
    }
}

// per-frame {min,max} of the four DataSet signals (fft.hpp:229-232): grid = (frames, 4)
__global__ void ranges_kernel(const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2, const float* __restrict__ s3, int nout,
                              float* __restrict__ ranges) {
    __shared__ float smin[256], smax[256];
    const float*     base[4] = {s0, s1, s2, s3};
    const float*     s       = base[blockIdx.y];
    float            mn = FLT_MAX, mx = -FLT_MAX;
    if (s) {
        s += (long)blockIdx.x * nout;
        for (int k = threadIdx.x; k < nout; k += blockDim.x) { const float v = s[k]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    }
    smin[threadIdx.x] = mn;
    smax[threadIdx.x] = mx;
    __syncthreads();
    for (int off = blockDim.x / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + off]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ranges[((long)blockIdx.x * 4 + blockIdx.y) * 2 + 0] = s ? smin[0] : 0.f;
        ranges[((long)blockIdx.x * 4 + blockIdx.y) * 2 + 1] = s ? smax[0] : 0.f;
    }
}

} // namespace gr4

using namespace gr4;

namespace gr4 { // chain_fused.hip
struct ChainFused;
int  chain_fused_create(ChainFused** out, const float* taps, size_t ntaps, size_t fft_size, int window, int algo);
int  chain_fused_fft_mag2(ChainFused* c, const float* d_in, size_t n_frames, float* d_mag2, const float* d_window, hipStream_t st);
int  chain_fused_fft_spectrum(ChainFused* c, const float* d_in, size_t n_frames, float* d_spectrum, const float* d_window, hipStream_t st);
void chain_fused_destroy(ChainFused* c);
} // namespace gr4

// A power-of-two transform of any supported length as a building block: frames in, every requested output of the natural-order spectrum out.
//   M <= 8192:            one launch of the block kernels
//   M = N1 * 4096 <= 2^20: the four-step pipeline above through the engine's own scratch buffers (N1 <= 16: column transforms in registers)
struct Pow2Engine {
    size_t       M  = 0;
    int          n1 = 0; // 0: single launch
    FftPlanDev   plan{}, plan_cols{};
    DeviceBuffer tw /*W_M^j*/, tw_rows /*W_4096^j*/, tw_cols /*W_N1^j, N1 > 16*/, s0, s1;
    DeviceBuffer tw_grid; // M = 65536: W_M^{n2 k1} as a table [k1][n2] (256 x 256): the column kernel reads it in the same 128-byte segments as the samples
};
constexpr size_t kFftMaxPow2 = size_t(1) << 20;      // largest power-of-two transform (N1 = 256)
constexpr long   kFftBatchElems = 1L << 25;          // multi-kernel paths: complex elements per scratch buffer (256 MiB); longer inputs run in batches of frames

struct gr4hip_fft {
    int          in_dtype = GR4HIP_C32;
    size_t       N        = 0;
    int          window   = GR4HIP_WIN_HANN;
    int          flags    = 0;
    FftPlanDev   plan{};
    DeviceBuffer d_window, d_tw, d_phase_raw;
    // multi-kernel paths: kind 1 = N1 * 4096 four-step, kind 2 = Bluestein with M-point transforms
    int          kind = 0;
    size_t       M    = 0;
    Pow2Engine   eng;  // the power-of-two transform both are built on (kind 1: of N points, kind 2: of M points)
    DeviceBuffer d_chirp, d_chirpF, d_scratchA, d_scratchB;
    gr4::ChainFused* pipe = nullptr; // N = 8192 complex, |X|^2 only: the persistent frame pipeline of chain_fused.hip (built on first use)
    gr4hip_ewise*    post = nullptr; // gr4hip_fft_set_epilogue: float blocks behind |X|^2, in the transform's launch
    ~gr4hip_fft();
};

namespace gr4 {
// shared with chain.hip
int fft_build_plan(size_t N, FftPlanDev* plan) {
    plan->smooth = 0;
    if (!is_pow2(N) && fft_is_smooth235(N) && N <= 8192) { // SimdFFT's radix-3 / radix-5 sizes (SimdFFT.hpp:348-375): mixed-radix passes, one launch
        plan->smooth = 1;
        return fft_build_smooth_plan(N, plan);
    }
    if (!is_pow2(N) || N < 2) { set_error("fft: size %zu is not a power of two >= 2 (device path)", N); return GR4HIP_UNSUPPORTED; }
    if (N > 8192) { set_error("fft: size %zu exceeds the single-workgroup LDS path (max 8192)", N); return GR4HIP_UNSUPPORTED; }
    plan->N  = (int)N;
    int l    = ilog2(N);
    int np   = 0;
    while (l >= 3) { plan->radix[np++] = 8; l -= 3; }
    if (l == 2) plan->radix[np++] = 4;
    if (l == 1) plan->radix[np++] = 2;
    plan->npass = np;
    plan->tf    = N >= 8 ? (int)(N / 8) : 1;
    plan->fpb   = plan->tf >= 64 ? 1 : 64 / plan->tf;
    return GR4HIP_OK;
}

int fft_upload_twiddles(size_t N, DeviceBuffer* buf) {
    std::vector<float> tw(2 * N);
    for (size_t k = 0; k < N; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)N;
        tw[2 * k]     = (float)std::cos(a);
        tw[2 * k + 1] = (float)std::sin(a);
    }
    int rc = buf->ensure(tw.size() * sizeof(float));
    if (rc) return rc;
    GR4_HIP_TRY(upload_fresh(buf->ptr, tw.data(), tw.size() * sizeof(float)));
    return GR4HIP_OK;
}

int fft_launch(const FftPlanDev& plan, const float* d_in, const float* d_window, const float2* d_tw, const FftOutputs& o, long n_frames, hipStream_t st) {
    if (plan.smooth) return fft_smooth_launch(plan, d_in, d_window, d_tw, o, n_frames, st);
    if (o.real_input && !o.spectrum && !o.mag2) { // real frames of 2N samples as N complex points + split (half the butterflies)
        switch (plan.N) {
        case 512: return fft_fast_launch<8, true>(d_in, d_window, d_tw, o, n_frames, st);
        case 1024: return fft_fast_launch<9, true>(d_in, d_window, d_tw, o, n_frames, st);
        case 2048: return fft_fast_launch<10, true>(d_in, d_window, d_tw, o, n_frames, st);
        case 4096: return fft_fast_launch<11, true>(d_in, d_window, d_tw, o, n_frames, st);
        case 8192: return fft_fast_launch<12, true>(d_in, d_window, d_tw, o, n_frames, st);
        default: break;
        }
    }
    switch (plan.N) { // compile-time plans; everything else takes the generic radix-8/4/2 kernel
    case 256: return fft_fast_launch_256(d_in, d_window, d_tw, o, n_frames, st);
    case 512: return fft_fast_launch<9>(d_in, d_window, d_tw, o, n_frames, st);
    case 1024: return fft_fast_launch<10>(d_in, d_window, d_tw, o, n_frames, st);
    case 2048: return fft_fast_launch<11>(d_in, d_window, d_tw, o, n_frames, st);
    case 4096: return fft_fast_launch<12>(d_in, d_window, d_tw, o, n_frames, st);
    case 8192: return fft_fast_launch_8192(d_in, d_window, d_tw, o, n_frames, st);
    default: break;
    }
    const size_t lds  = (size_t)plan.fpb * plan.N * sizeof(float2);
    const int    bs   = plan.tf * plan.fpb;
    const long   grid = ceil_div(n_frames, (long)plan.fpb);
    if (lds > 48 * 1024) GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fft_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fft_block_kernel, dim3((unsigned)grid), dim3(bs), lds, st, d_in, d_window, d_tw, plan, o, n_frames);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}
} // namespace gr4

static int engine_create(Pow2Engine* e, size_t M) {
    e->M   = M;
    int rc = fft_upload_twiddles(M, &e->tw);
    if (rc) return rc;
    if (M <= 8192) return fft_build_plan(M, &e->plan);
    e->n1 = (int)(M / 4096);
    rc    = fft_upload_twiddles(4096, &e->tw_rows);
    if (!rc && M == 65536) {
        std::vector<float> g((size_t)2 * 65536);
        for (int k1 = 0; k1 < 256; ++k1)
            for (int n2 = 0; n2 < 256; ++n2) {
                const double ang = -2.0 * M_PI * (double)((k1 * n2) & 65535) / 65536.0;
                g[2 * ((size_t)k1 * 256 + n2)]     = (float)std::cos(ang);
                g[2 * ((size_t)k1 * 256 + n2) + 1] = (float)std::sin(ang);
            }
        rc = e->tw_grid.ensure(g.size() * sizeof(float));
        if (!rc) GR4_HIP_TRY(upload_fresh(e->tw_grid.ptr, g.data(), g.size() * sizeof(float)));
    }
    if (!rc && e->n1 > 16) {
        rc = fft_build_plan((size_t)e->n1, &e->plan_cols);
        if (!rc) rc = fft_upload_twiddles((size_t)e->n1, &e->tw_cols);
    }
    return rc;
}

// n_frames frames of M points (complex, or real when fin.real_input; `window` may be null) -> the outputs named in `fin`, written from frame index frame0 on
static int engine_run(Pow2Engine* e, const float* d_in, const float* window, const FftOutputs& fin, long n_frames, long frame0, hipStream_t st) {
    const long M = (long)e->M;
    if (e->n1 == 0) {
        if (frame0 != 0) { set_error("fft: internal: a single-launch transform has no frame offset"); return GR4HIP_RUNTIME_ERROR; }
        return fft_launch(e->plan, d_in, window, static_cast<const float2*>(e->tw.ptr), fin, n_frames, st);
    }
    const int  n1    = e->n1;
    const long total = n_frames * M;
    int        rc    = e->s0.ensure((size_t)total * sizeof(float2));
    if (!rc) rc = e->s1.ensure((size_t)total * sizeof(float2));
    if (rc) return rc;
    float2*    s0  = static_cast<float2*>(e->s0.ptr);
    float2*    s1  = static_cast<float2*>(e->s1.ptr);
    const auto twN = static_cast<const float2*>(e->tw.ptr);
    FftOutputs spec{};
    if (M == 65536 && !dev_switch(kDevFftFourStep64k)) { // 256 x 256 in two kernels (32 B of HBM traffic per point instead of 48)
        const auto tw4096 = static_cast<const float2*>(e->tw_rows.ptr);
        hipLaunchKernelGGL(fft_64k_cols_kernel, dim3((unsigned)(n_frames * 16)), dim3(256), 0, st, d_in, window, static_cast<const float2*>(e->tw_grid.ptr), tw4096, s0, fin.real_input);
        GR4_LAUNCH_CHECK();
        hipLaunchKernelGGL(fft_64k_rows_kernel, dim3((unsigned)(n_frames * 16)), dim3(256), 0, st, (const float2*)s0, tw4096, fin, frame0);
        GR4_LAUNCH_CHECK();
        return GR4HIP_OK;
    }
    if (n1 <= 16) {
        const dim3 grid((unsigned)ceil_div(n_frames * 4096L, 256L));
        if (n1 == 4) hipLaunchKernelGGL(fft_big_cols_kernel<4>, grid, dim3(256), 0, st, d_in, window, twN, s0, n_frames, fin.real_input);
        else if (n1 == 8) hipLaunchKernelGGL(fft_big_cols_kernel<8>, grid, dim3(256), 0, st, d_in, window, twN, s0, n_frames, fin.real_input);
        else hipLaunchKernelGGL(fft_big_cols_kernel<16>, grid, dim3(256), 0, st, d_in, window, twN, s0, n_frames, fin.real_input);
        GR4_LAUNCH_CHECK();
    } else {
        const dim3 tiles((unsigned)(n_frames * 128L * (n1 / 32)));
        hipLaunchKernelGGL(fft_big_gather_kernel, tiles, dim3(256), 0, st, d_in, window, s0, n1, fin.real_input);
        GR4_LAUNCH_CHECK();
        spec.spectrum = reinterpret_cast<float*>(s1);
        rc            = fft_launch(e->plan_cols, reinterpret_cast<const float*>(s0), nullptr, static_cast<const float2*>(e->tw_cols.ptr), spec, n_frames * 4096L, st);
        if (rc) return rc;
        hipLaunchKernelGGL(fft_big_scatter_kernel, tiles, dim3(256), 0, st, (const float2*)s1, twN, s0, n1);
        GR4_LAUNCH_CHECK();
    }
    spec.spectrum = reinterpret_cast<float*>(s1);
    rc            = fft_fast_launch<12>(reinterpret_cast<const float*>(s0), nullptr, static_cast<const float2*>(e->tw_rows.ptr), spec, n_frames * n1, st);
    if (rc) return rc;
    if (n1 <= 16) hipLaunchKernelGGL(fft_emit_kernel, dim3((unsigned)ceil_div(total, 256L)), dim3(256), 0, st, (const float2*)s1, fin, (int)M, n1, n_frames, frame0);
    else hipLaunchKernelGGL(fft_big_emit_kernel, dim3((unsigned)(n_frames * 128L * (n1 / 32))), dim3(256), 0, st, (const float2*)s1, fin, n1, frame0);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

static int fft_upload_bluestein(size_t N, size_t M, DeviceBuffer* d_cconj, DeviceBuffer* d_Bf);

extern "C" {

int gr4hip_fft_create(gr4hip_fft_t** out, int in_dtype, size_t fft_size, int window, int flags) {
    GR4_REQUIRE(out, "fft: null output handle");
    GR4_REQUIRE(in_dtype == GR4HIP_F32 || in_dtype == GR4HIP_C32, "fft: input dtype must be F32 or C32 (got %d)", in_dtype);
    GR4_REQUIRE(window >= GR4HIP_WIN_NONE && window <= GR4HIP_WIN_KAISER, "fft: unknown window %d", window);
    auto* f = new (std::nothrow) gr4hip_fft();
    GR4_REQUIRE(f, "out of host memory");
    f->in_dtype = in_dtype;
    f->N        = fft_size;
    f->window   = window;
    f->flags    = flags;
    int rc = GR4HIP_OK;
    if ((is_pow2(fft_size) || fft_is_smooth235(fft_size)) && fft_size >= 2 && fft_size <= 8192) {
        rc = fft_build_plan(fft_size, &f->plan);
        if (!rc) rc = fft_upload_twiddles(fft_size, &f->d_tw);
    } else if (is_pow2(fft_size) && fft_size <= kFftMaxPow2) { // 16384 .. 2^20: four-step with 4096-point rows
        f->kind = 1;
        f->M    = fft_size;
        rc      = engine_create(&f->eng, fft_size);
    } else if (fft_size >= 2 && 2 * fft_size - 1 <= kFftMaxPow2) { // every other size (SimdFFT's radix-3/5 sizes included): Bluestein
        f->kind = 2;
        size_t M = 1;
        while (M < 2 * fft_size - 1) M <<= 1;
        f->M = M;
        rc   = engine_create(&f->eng, M);
        if (!rc) rc = fft_upload_bluestein(fft_size, M, &f->d_chirp, &f->d_chirpF);
        if (!rc && M <= 8192) { // the fused single-launch form
            rc = fft_build_plan(M, &f->plan);
            if (!rc) rc = fft_upload_twiddles(M, &f->d_tw);
        }
    } else {
        set_error("fft: size %zu is outside the device paths (powers of two <= %zu, any size <= %zu)", fft_size, kFftMaxPow2, kFftMaxPow2 / 2);
        rc = GR4HIP_UNSUPPORTED;
    }
    if (!rc && window != GR4HIP_WIN_NONE && window != GR4HIP_WIN_RECTANGULAR) {
        std::vector<float> w(fft_size);
        rc = make_window(window, w.data(), fft_size, 1.6f); // fft.hpp:141: create(_window, _windowType) -> default beta
        if (!rc) rc = f->d_window.ensure(fft_size * sizeof(float));
        if (!rc) { hipError_t e = upload_fresh(f->d_window.ptr, w.data(), fft_size * sizeof(float)); if (e != hipSuccess) { set_error("window upload: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
    }
    if (rc) { delete f; return rc; }
    *out = f;
    return GR4HIP_OK;
}

// unwrap and ranges passes behind the transform kernels
static int fft_finish(gr4hip_fft_t* f, const FftOutputs& o, float* d_phase_final, float* d_ranges, size_t n_frames, int nout, bool unwrap, bool fused_ranges, hipStream_t st) {
    (void)f;
    if (unwrap) {
        const int bs = nout >= 256 ? 256 : 64;
        hipLaunchKernelGGL(unwrap_kernel, dim3((unsigned)n_frames), dim3(bs), bs * sizeof(int), st, (const float*)o.phase_raw, d_phase_final, nout, o.in_deg,
                           o.real_input ? 0 : 1);
        GR4_LAUNCH_CHECK();
    }
    if (d_ranges && !fused_ranges) {
        hipLaunchKernelGGL(ranges_kernel, dim3((unsigned)n_frames, 4), dim3(256), 0, st, (const float*)o.mag, (const float*)d_phase_final, (const float*)o.re,
                           (const float*)o.im, nout, d_ranges);
        GR4_LAUNCH_CHECK();
    }
    return GR4HIP_OK;
}

// chirp tables of the Bluestein path: cconj[n] = e^{-i pi n^2 / N} (n < N) and the M-point transform of the wrapped chirp c[+-n]
static int fft_upload_bluestein(size_t N, size_t M, DeviceBuffer* d_cconj, DeviceBuffer* d_Bf) {
    std::vector<float>                cc(2 * N);
    std::vector<std::complex<double>> b(M, {0.0, 0.0});
    for (size_t n = 0; n < N; ++n) {
        const double ang = M_PI * (double)((n * n) % (2 * N)) / (double)N; // n^2 mod 2N keeps the argument exact
        cc[2 * n]        = (float)std::cos(ang);
        cc[2 * n + 1]    = (float)-std::sin(ang);
        const std::complex<double> c(std::cos(ang), std::sin(ang));
        b[n] = c;
        if (n) b[M - n] = c;
    }
    // FFT_M(b) in double: iterative radix-2 (M is a power of two), bit reversal first
    for (size_t i = 1, j = 0; i < M; ++i) {
        size_t bit = M >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(b[i], b[j]);
    }
    for (size_t len = 2; len <= M; len <<= 1) {
        const double ang = -2.0 * M_PI / (double)len;
        for (size_t i = 0; i < M; i += len)
            for (size_t k = 0; k < len / 2; ++k) {
                const std::complex<double> w(std::cos(ang * (double)k), std::sin(ang * (double)k));
                const std::complex<double> u = b[i + k], v = b[i + k + len / 2] * w;
                b[i + k]           = u + v;
                b[i + k + len / 2] = u - v;
            }
    }
    std::vector<float> bf(2 * M);
    for (size_t k = 0; k < M; ++k) { bf[2 * k] = (float)b[k].real(); bf[2 * k + 1] = (float)b[k].imag(); }
    int rc = d_cconj->ensure(cc.size() * sizeof(float));
    if (!rc) rc = d_Bf->ensure(bf.size() * sizeof(float));
    if (rc) return rc;
    GR4_HIP_TRY(upload_fresh(d_cconj->ptr, cc.data(), cc.size() * sizeof(float)));
    GR4_HIP_TRY(upload_fresh(d_Bf->ptr, bf.data(), bf.size() * sizeof(float)));
    return GR4HIP_OK;
}

// the multi-kernel paths (kind 1: four-step, kind 2: Bluestein); `o` already carries the output pointers and flags.  The scratch buffers are bounded:
// long inputs run in batches of frames (the emitting kernels address the outputs with the absolute frame index)
static int fft_run_multi(gr4hip_fft_t* f, const float* d_in, long n_frames, const FftOutputs& o, hipStream_t st) {
    const long   N      = (long)f->N, M = (long)f->M;
    const float* win    = static_cast<const float*>(f->d_window.ptr);
    const long   batch  = std::max(1L, kFftBatchElems / M);
    const long   in_per = o.real_input ? N : 2 * N; // floats per input frame
    if (f->kind == 2 && M >= 256 && M <= 4096 && !dev_switch(kDevFftBluesteinPipeline) && !dev_switch(kDevFftBluesteinGeneric)) { // the compile-time 16 x 16 x R3 plan, register to register
        const auto cc = static_cast<const float2*>(f->d_chirp.ptr), bf = static_cast<const float2*>(f->d_chirpF.ptr), tw = static_cast<const float2*>(f->d_tw.ptr);
        switch (M) {
        case 256: return bluestein_fast_launch<8>(d_in, win, cc, bf, tw, o, (int)N, n_frames, st);
        case 512: return bluestein_fast_launch<9>(d_in, win, cc, bf, tw, o, (int)N, n_frames, st);
        case 1024: return bluestein_fast_launch<10>(d_in, win, cc, bf, tw, o, (int)N, n_frames, st);
        case 2048: return bluestein_fast_launch<11>(d_in, win, cc, bf, tw, o, (int)N, n_frames, st);
        default: return bluestein_fast_launch<12>(d_in, win, cc, bf, tw, o, (int)N, n_frames, st);
        }
    }
    if (f->kind == 2 && M <= 8192 && !dev_switch(kDevFftBluesteinPipeline)) { // (developer switch: the five-kernel pipeline, which the tests compare)
        const size_t lds = (size_t)f->plan.fpb * M * sizeof(float2);
        if (lds > 48 * 1024) GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(bluestein_fused_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(bluestein_fused_kernel, dim3((unsigned)ceil_div(n_frames, (long)f->plan.fpb)), dim3(f->plan.tf * f->plan.fpb), lds, st, d_in, win,
                           static_cast<const float2*>(f->d_chirp.ptr), static_cast<const float2*>(f->d_chirpF.ptr), static_cast<const float2*>(f->d_tw.ptr), f->plan, o, (int)N, n_frames);
        GR4_LAUNCH_CHECK();
        return GR4HIP_OK;
    }
    for (long f0 = 0; f0 < n_frames; f0 += batch) {
        const long   nf = std::min(batch, n_frames - f0);
        const float* in = d_in + f0 * in_per;
        if (f->kind == 1) {
            int rc = engine_run(&f->eng, in, win, o, nf, f0, st);
            if (rc) return rc;
            continue;
        }
        const long total = nf * M;
        int        rc    = f->d_scratchA.ensure((size_t)total * sizeof(float2));
        if (!rc) rc = f->d_scratchB.ensure((size_t)total * sizeof(float2));
        if (rc) return rc;
        float2*    a  = static_cast<float2*>(f->d_scratchA.ptr);
        float2*    b  = static_cast<float2*>(f->d_scratchB.ptr);
        const auto cc = static_cast<const float2*>(f->d_chirp.ptr);
        hipLaunchKernelGGL(bluestein_pre_kernel, dim3((unsigned)ceil_div(total, 256L)), dim3(256), 0, st, in, win, cc, a, (int)N, (int)M, nf, o.real_input);
        GR4_LAUNCH_CHECK();
        FftOutputs spec{};
        spec.spectrum = reinterpret_cast<float*>(b);
        rc            = engine_run(&f->eng, reinterpret_cast<const float*>(a), nullptr, spec, nf, 0, st);
        if (rc) return rc;
        hipLaunchKernelGGL(bluestein_mul_kernel, dim3((unsigned)ceil_div(total, 256L)), dim3(256), 0, st, b, (const float2*)f->d_chirpF.ptr, (int)M, total);
        GR4_LAUNCH_CHECK();
        spec.spectrum = reinterpret_cast<float*>(a);
        rc            = engine_run(&f->eng, reinterpret_cast<const float*>(b), nullptr, spec, nf, 0, st);
        if (rc) return rc;
        hipLaunchKernelGGL(bluestein_post_kernel, dim3((unsigned)ceil_div(nf * N, 256L)), dim3(256), 0, st, (const float2*)a, cc, o, (int)N, (int)M, nf, f0);
        GR4_LAUNCH_CHECK();
    }
    return GR4HIP_OK;
}

static int fft_run(gr4hip_fft_t* f, const void* d_in, size_t n_frames, FftOutputs o, float* d_phase_final, float* d_ranges, gr4hip_stream_t stream) {
    GR4_REQUIRE(f, "fft: null handle");
    if (n_frames == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in, "fft: null input");
    hipStream_t st   = as_stream(stream);
    o.real_input     = f->in_dtype == GR4HIP_F32;
    o.in_db          = (f->flags & GR4HIP_FFT_OUTPUT_IN_DB) != 0;
    o.in_deg         = (f->flags & GR4HIP_FFT_OUTPUT_IN_DEG) != 0;
    const int  nout  = o.real_input ? (int)f->N / 2 : (int)f->N;
    const bool unwrap = (f->flags & GR4HIP_FFT_UNWRAP_PHASE) && d_phase_final;
    if (unwrap) {
        int rc = f->d_phase_raw.ensure(n_frames * nout * sizeof(float));
        if (rc) return rc;
        o.phase_raw = static_cast<float*>(f->d_phase_raw.ptr);
        o.phase     = nullptr;
    } else {
        o.phase     = d_phase_final;
        o.phase_raw = nullptr;
    }
    if (f->kind != 0) {
        int rc = fft_run_multi(f, static_cast<const float*>(d_in), (long)n_frames, o, st);
        if (rc) return rc;
        return fft_finish(f, o, d_phase_final, d_ranges, n_frames, nout, unwrap, false, st);
    }
    const bool fast         = f->N >= 256 && f->N <= 8192 && !f->plan.smooth; // fft_fast_kernel sizes (powers of two; the mixed-radix kernel leaves the ranges to ranges_kernel)
    const bool fused_ranges = d_ranges && fast && !unwrap;  // the unwrapped phase only exists after unwrap_kernel
    o.ranges                = fused_ranges ? d_ranges : nullptr;
    int rc = fft_launch(f->plan, static_cast<const float*>(d_in), static_cast<const float*>(f->d_window.ptr), static_cast<const float2*>(f->d_tw.ptr), o,
                        (long)n_frames, st);
    if (rc) return rc;
    return fft_finish(f, o, d_phase_final, d_ranges, n_frames, nout, unwrap, fused_ranges, st);
}

int gr4hip_fft_process(gr4hip_fft_t* f, const void* d_in, size_t n_frames, float* d_mag, float* d_phase, float* d_re, float* d_im, float* d_ranges,
                       gr4hip_stream_t stream) {
    FftOutputs o{};
    o.mag = d_mag;
    o.re  = d_re;
    o.im  = d_im;
    return fft_run(f, d_in, n_frames, o, d_phase, d_ranges, stream);
}

// 8192-point complex frames, >= one frame per CU: |X|^2 and the raw spectrum run on the frame pipeline of the fused chain kernel without its filter
// (LDS-DMA prefetch of the next frame during the transform of this one); everything else goes to the FFT block kernels
static bool fft_on_frame_pipeline(const gr4hip_fft_t* f, size_t n_frames) {
    return f->N == 8192 && f->in_dtype == GR4HIP_C32 && f->kind == 0 && n_frames >= 256 && !dev_switch(kDevFftNoPipeline);
}
static int fft_pipe(gr4hip_fft_t* f) {
    if (f->pipe) return GR4HIP_OK;
    const float one = 1.f;
    return gr4::chain_fused_create(&f->pipe, &one, 1, 8192, GR4HIP_WIN_NONE, GR4HIP_CHAIN_FUSED_FD);
}

int gr4hip_fft_spectrum(gr4hip_fft_t* f, const void* d_in, size_t n_frames, float* d_spectrum, gr4hip_stream_t stream) {
    GR4_REQUIRE(d_spectrum || n_frames == 0, "fft_spectrum: null output");
    GR4_REQUIRE(f, "fft_spectrum: null handle");
    if (fft_on_frame_pipeline(f, n_frames)) {
        int rc = fft_pipe(f);
        if (rc) return rc;
        const bool windowed = f->window != GR4HIP_WIN_NONE && f->window != GR4HIP_WIN_RECTANGULAR;
        return gr4::chain_fused_fft_spectrum(f->pipe, static_cast<const float*>(d_in), n_frames, d_spectrum, windowed ? static_cast<const float*>(f->d_window.ptr) : nullptr, as_stream(stream));
    }
    FftOutputs o{};
    o.spectrum = d_spectrum;
    return fft_run(f, d_in, n_frames, o, nullptr, nullptr, stream);
}

gr4hip_fft::~gr4hip_fft() { if (pipe) gr4::chain_fused_destroy(pipe); delete post; }

int gr4hip_fft_set_epilogue(gr4hip_fft_t* f, const gr4hip_ewise_t* prog) {
    GR4_REQUIRE(f, "fft_set_epilogue: null handle");
    if (prog && prog->dtype != GR4HIP_F32) { set_error("fft_set_epilogue: |X|^2 is a float stream, the program's dtype is %d", prog->dtype); return GR4HIP_UNSUPPORTED; }
    gr4hip_ewise* copy = nullptr;
    if (prog && !prog->user.empty()) {
        copy = ewise_clone(prog);
        GR4_REQUIRE(copy, "out of host memory");
    }
    delete f->post;
    f->post = copy;
    return GR4HIP_OK;
}

int gr4hip_fft_mag2(gr4hip_fft_t* f, const void* d_in, size_t n_frames, float* d_mag2, gr4hip_stream_t stream) {
    GR4_REQUIRE(d_mag2 || n_frames == 0, "fft_mag2: null output");
    GR4_REQUIRE(f, "fft_mag2: null handle");
    EwiseHook post;
    if (f->post) { if (const int rc = ewise_device_ops(f->post, &post, as_stream(stream))) return rc; }
    // 8192-point complex frames, >= one frame per CU: the frame pipeline of the fused chain kernel without its filter (LDS-DMA prefetch of the next
    // frame during the transform of this one); everything else goes to the FFT block kernels
    if (fft_on_frame_pipeline(f, n_frames)) {
        int rc = fft_pipe(f);
        if (rc) return rc;
        const bool windowed = f->window != GR4HIP_WIN_NONE && f->window != GR4HIP_WIN_RECTANGULAR;
        rc = gr4::chain_fused_fft_mag2(f->pipe, static_cast<const float*>(d_in), n_frames, d_mag2, windowed ? static_cast<const float*>(f->d_window.ptr) : nullptr, as_stream(stream));
        // (the frame pipeline has no store hook: an epilogue is one element-wise launch over its output, in place)
        if (!rc && post.n_ops > 0) rc = ewise_run(post, GR4HIP_F32, d_mag2, d_mag2, (long)(n_frames * f->N), as_stream(stream));
        return rc;
    }
    FftOutputs o{};
    o.mag2      = d_mag2;
    o.mag2_post = post; // every kernel of this file stores |X|^2 through emit_bin or the fast kernel's register epilogue: the program rides in the transform's launch
    return fft_run(f, d_in, n_frames, o, nullptr, nullptr, stream);
}

int gr4hip_fft_destroy(gr4hip_fft_t* f) { delete f; return GR4HIP_OK; }

int gr4hip_fft_plan(size_t fft_size, int* kind, int* radices, int* n_passes) {
    GR4_REQUIRE(kind && radices && n_passes, "fft_plan: null argument");
    *n_passes = 0;
    if (is_pow2(fft_size) && fft_size >= 2 && fft_size <= 8192) { *kind = 0; return GR4HIP_OK; }
    if (fft_is_smooth235(fft_size) && fft_size <= 8192) {
        FftPlanDev plan{};
        int rc = fft_build_smooth_plan(fft_size, &plan);
        if (rc) return rc;
        *kind = 3;
        *n_passes = plan.npass;
        for (int i = 0; i < plan.npass; ++i) radices[i] = plan.radix[i];
        return GR4HIP_OK;
    }
    if (is_pow2(fft_size) && fft_size <= kFftMaxPow2) { *kind = 1; return GR4HIP_OK; }
    if (fft_size >= 2 && 2 * fft_size - 1 <= kFftMaxPow2) { *kind = 2; return GR4HIP_OK; }
    set_error("fft: size %zu is outside the device paths", fft_size);
    return GR4HIP_UNSUPPORTED;
}

} // extern "C"

