// fir_decim_fd.hip -- decimate-by-8 real FIR (<= 1025 taps) in the frequency domain for gfx950: BASELINE.json configs[2]'s filter.
//
// BasicFilterProto's decimating processBulk (blocks/filter/.../time_domain_filter.hpp:190-204) keeps y[8 m] of y[n] = sum_k b[k] x[n - k].  In direct
// (polyphase) form that is 2 K / 8 = 256 flop per input sample at 1024 taps -- FP32-bound at ~340 G input samples/s on the matrix pipe (fir_batched.hip).
// Here the kept outputs come from overlap-save blocks of N = 8192 real samples (overlap V = 1024, hop 7168 = 896 outputs):
//
//   z[n] = xb[2 n] + i xb[2 n + 1]                       the real block as 4096 complex points
//   Z    = FFT_4096(z)                                     decimation in frequency 8 x 512: cross pass + wave-private 512-point transforms (chain16.hip)
//   G[r] = sum_{j<4} R[r + 1024 j] Z[r + 1024 j]          r < 1024: real-FFT untangling, times H, and the alias sum of the decimation in ONE table:
//          R[k] = H[k] (1 - i W_8192^k)/2 + conj( H[4096-k] (1 + i W_8192^{4096-k})/2 ),  R[0] := (R[0] + R[4096])/2;  all four terms sit in ONE lane
//   y[8 i] = Re( IFFT_1024(G) )[i] / 4                    valid for i >= 128: wave-private 128-point inverse transforms + one radix-8 pass across the waves
//
// (derivation and a float64 check of the identity: tools/proto_decim_fd.py).  ~35 VALU lane-operations per input sample instead of 256 flop; HBM 4 B in
// (+ 14 % overlap re-read) + 0.5 B out per sample.  One workgroup of 512 lanes (8 waves, <= 128 VGPRs) per block, two workgroups per CU, persistent,
// the next block streams into the landing buffer by LDS-DMA while this one is transformed.
//
// Where the time goes (timing-only builds, 2^27 input samples, tools/abdf.sh; the full kernel: 0.216-0.224 ms): without the landing DMA 0.158-0.162, with the
// DMA re-reading a block that sits in the L2 0.175, without the wave-private 512-point transforms 0.149, without the inverse transforms 0.194, neither (DMA,
// cross pass, product, final pass: the memory skeleton) 0.128.  Memory and arithmetic do not hide behind each other -- as for the fused chain
// the sum behaves like an energy budget, not like max(compute, memory).  Warming the L2 one, two or three blocks ahead with 4-byte LDS-DMA touches measured
// 0 / -5 / -8 %: the traffic itself costs, not its latency.
#include "common.hpp"
#include "buffer_ops.hpp"
#include "fft_radix.hpp"
#include "wave16_common.hpp"

#include <cmath>
#include <cstring>
#include <complex>

namespace gr4 {

constexpr int kDfN   = 8192;            // real samples per block
constexpr int kDfV   = 1024;            // overlap (history) samples
constexpr int kDfHop = kDfN - kDfV;     // 7168 new samples = 896 outputs per block
constexpr int kDfT   = 512;             // lanes per workgroup
constexpr int kDfRS  = 4552;            // bytes per wave-private region (568 float2 + 8)
constexpr int kDfOffW = 32768;          // landing buffer: 8192 floats at 0
constexpr int kDfOffF = kDfOffW + 8 * kDfRS; // F_q[i''] of the inverse transforms: 8 x 128 float2 (pitch 129: the final pass reads 8 rows per lane)
constexpr int kDfLds  = kDfOffF + 8 * 129 * 8;
constexpr int kDfLdsAll = kDfLds + 64; // + the 16 verdict words of the dynamic-range guard (two slots x 8 waves)
static_assert(2 * kDfLdsAll <= 160 * 1024, "two workgroups per CU");
constexpr float kDecimFdBlockThreshold = 2.5e-3f; // = kDecimFdMinPowerRatio of fir.hip: the guard's output / input power threshold, applied to every block by itself

struct DecimFdArgs {
    const float*  x;       // input samples (the span); block j covers positions j * 7168 - 1024 .. + 8191
    const float*  hist;    // 1024 samples in front of the span
    const float2* twX;     // [512][8]  W_4096^{m q}
    const float2* tw1;     // [64][8]   W_512^{l k}
    const float2* tw2;     // [8][8]    W_64^{l0 k}
    const float2* R;       // [8][64][8] R[8 k' + q] (normalisation folded in), k' = c16_out_bin(lane, reg)
    const float2* twI1;    // [16]   W_128^{-k_lo}   (the inverse passes build their few powers from one exact base per lane: 4 registers instead of 32)
    const float2* twI2;    // [128]  W_1024^{-i''}
    float*        y;       // 896 outputs per block
    long          n_blocks;
    // a span that is not a whole number of hops: its last, partial block (index tail_blk = n_blocks - 1) is read from a zero-padded staging image of 8192 samples
    // (FIR causality: the padding cannot reach the valid outputs) and only its first tail_out outputs are stored.  tail_blk < 0: no such block
    const float*  x_tail;
    long          tail_blk;
    int           tail_out;
    float*        pw;      // dynamic-range guard (optional): 16 slots of {sum x^2, sum y^2} over EVERY block (8192 inputs / <= 896 outputs each), then the workgroups-done counter and the flag word
    float*        pw_host; // page-locked, device-mapped {in, out, sequence number}: written by the last workgroup to finish
    unsigned      pw_seq;
    float         pw_thr;  // a block whose output power is below pw_thr x its input power marks the launch (word 33 of pw, word 3 of pw_host)
    // fir_decim_fd_kernel<true>: the biquad cascade behind the decimator (Filter<float>::processOne over the sections, FilterTool.hpp:244-246) in the SAME launch --
    // BASELINE configs[2] without the decimated stream in HBM.  y then receives the cascade's output.
    const float*  iir;           // [iir_nsec][32]: b0 b1 b2 a1 a2 - - - | P_k (k < 5): A^(14 2^k), row-major 2 x 2, A = [[-a1, -a2], [1, 0]] (the section's state transition)
    const float*  iir_state;     // [4][2]: the cascade's direct-form-II delay lines (w[n-1], w[n-2]) in front of the span
    float*        iir_state_out; // ... behind it (written by the workgroup that owns the span's last block)
    int           iir_nsec, run_blocks, warm_blocks; // contiguous blocks per workgroup; blocks every run but the first starts early from zero state (the cascade's memory fades: ||Phi_896^warm|| <= 1e-8)
};
constexpr int kDfLdsIir = kDfLdsAll + 896 * 4 + 8 * 4; // + the block's decimated samples on their way from the final pass (lanes 0 .. 127) to the cascade's wave
static_assert(2 * kDfLdsIir <= 160 * 1024, "two workgroups per CU, with the cascade");

// ---- the cascade on one block: 896 samples, ONE wave, lane l owns samples 14 l .. 14 l + 13 (in registers, filtered in place section by section).
// Per section: a run from zero state (lane 0: from the carried state) gives every chunk's end state; a Kogge-Stone scan inside each row of 16 lanes (DPP row
// shifts, the 2 x 2 powers A^(14 2^k) as uniform multipliers), the four row totals chained in scalar registers, and A^(14 (j + 1)) W_row by binary powers give
// every chunk's TRUE end state; shifted one lane up it is the next chunk's start state, from which the chunk is run again, now with outputs.
template <int CTRL>
__device__ __forceinline__ float decim_dpp(float v) { return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true)); }
__device__ __forceinline__ float decim_lane(float v, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), l)); }
typedef const __attribute__((address_space(4))) float* decim_ctab_t;
// (z0, z1): every chunk's end state from zero state (lane 0: from the carried state) -> every chunk's TRUE start state (c0, c1)
__device__ __forceinline__ void decim_iir_scan(decim_ctab_t tab, float z0, float z1, float T0, float T1, int lane, float& c0, float& c1) {
    const auto mul = [&](int k, float u0, float u1, float& r0, float& r1) { // (r0, r1) += P_k (u0, u1)
        const float p00 = tab[8 + 4 * k], p01 = tab[9 + 4 * k], p10 = tab[10 + 4 * k], p11 = tab[11 + 4 * k];
        r0 = fmaf(p00, u0, fmaf(p01, u1, r0));
        r1 = fmaf(p10, u0, fmaf(p11, u1, r1));
    };
    float s0 = z0, s1 = z1;
    { const float u0 = decim_dpp<0x111>(s0), u1 = decim_dpp<0x111>(s1); mul(0, u0, u1, s0, s1); } // row_shr:1 (lanes at a row's start read 0)
    { const float u0 = decim_dpp<0x112>(s0), u1 = decim_dpp<0x112>(s1); mul(1, u0, u1, s0, s1); } // row_shr:2
    { const float u0 = decim_dpp<0x114>(s0), u1 = decim_dpp<0x114>(s1); mul(2, u0, u1, s0, s1); } // row_shr:4
    { const float u0 = decim_dpp<0x118>(s0), u1 = decim_dpp<0x118>(s1); mul(3, u0, u1, s0, s1); } // row_shr:8
    // state at the start of row r from the rows before it: W_0 = 0, W_{r+1} = A^(14 16) W_r + (row r's total, at its lane 15)
    float w0 = 0.f, w1 = 0.f, W0 = 0.f, W1 = 0.f;
    const int row = lane >> 4;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        float n0 = decim_lane(s0, 16 * r + 15), n1 = decim_lane(s1, 16 * r + 15);
        mul(4, w0, w1, n0, n1);
        w0 = n0;
        w1 = n1;
        if (row == r + 1) { W0 = w0; W1 = w1; }
    }
    // + A^(14 (j + 1)) W, j = lane in row: the binary powers of j + 1 (1 .. 16)
    const int j1 = (lane & 15) + 1;
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        float h0 = 0.f, h1 = 0.f;
        mul(k, W0, W1, h0, h1);
        if ((j1 >> k) & 1) { W0 = h0; W1 = h1; }
    }
    s0 += W0;
    s1 += W1;
    // the chunk's start state: the end state of the chunk before it (wave_shr:1; lane 0: the carried state)
    c0 = decim_dpp<0x138>(s0);
    c1 = decim_dpp<0x138>(s1);
    if (lane == 0) { c0 = T0; c1 = T1; }
}
// the chunk's run from zero state (lane 0: from the carried state Ts): end state
__device__ __forceinline__ void decim_iir_first(const float (&x)[14], decim_ctab_t tab, const float* Ts, int lane, float& z0, float& z1) {
    const float a1 = tab[3], a2 = tab[4];
    z0 = lane == 0 ? Ts[0] : 0.f;
    z1 = lane == 0 ? Ts[1] : 0.f;
#pragma unroll
    for (int n = 0; n < 14; ++n) {
        const float w = fmaf(-a2, z1, fmaf(-a1, z0, x[n]));
        z1            = z0;
        z0            = w;
    }
}
// section `tab` on the chunk, in place: end state from zero state, scan, the run proper
__device__ __forceinline__ void decim_iir_section(float (&x)[14], decim_ctab_t tab, float* Ts, int lane) {
    const float a1 = tab[3], a2 = tab[4];
    float       z0, z1, c0, c1;
    decim_iir_first(x, tab, Ts, lane, z0, z1);
    decim_iir_scan(tab, z0, z1, Ts[0], Ts[1], lane, c0, c1);
    const float b0 = tab[0], b1 = tab[1], b2 = tab[2];
#pragma unroll
    for (int n = 0; n < 14; ++n) {
        const float w = fmaf(-a2, c1, fmaf(-a1, c0, x[n]));
        x[n]          = fmaf(b2, c1, fmaf(b1, c0, b0 * w));
        c1            = c0;
        c0            = w;
    }
    if (lane == 63) { Ts[0] = c0; Ts[1] = c1; } // the block's end state: what the next block starts from (one wave, LDS in program order)
}
__device__ __forceinline__ void ifft8(float2 (&v)[8]) { // inverse DFT-8: the forward butterfly on (im, re)
    float2 t[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) t[i] = make_float2(v[i].y, v[i].x);
    fft8(t);
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = make_float2(t[i].y, t[i].x);
}

// v[k] *= w^k, k = 1..7, powers by products of depth <= 3
__device__ __forceinline__ void mul_powers(float2 (&v)[8], float2 w) {
    const float2 w2 = cmul(w, w), w3 = cmul(w2, w), w4 = cmul(w2, w2);
    v[1] = cmul(v[1], w);
    v[2] = cmul(v[2], w2);
    v[3] = cmul(v[3], w3);
    v[4] = cmul(v[4], w4);
    v[5] = cmul(v[5], cmul(w4, w));
    v[6] = cmul(v[6], cmul(w4, w2));
    v[7] = cmul(v[7], cmul(w4, w3));
}

// landing: 32 one-KiB pieces per block, 4 per wave; the first 4 pieces of the span's first block come from the history
__device__ __forceinline__ void decim_dma(const DecimFdArgs& a, long blk, unsigned lds_L, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int    p   = 4 * wave + i;                       // piece: floats [256 p, 256 p + 256) of the block
        const long   pos = blk * kDfHop - kDfV + 256L * p;     // stream position of its first sample
        const float* src = blk == a.tail_blk ? a.x_tail + 256 * p : pos < 0 ? a.hist + (kDfV + pos) : a.x + pos; // (pieces never straddle position 0: 1024 = 4 pieces)
        dma16_1k(src + 4 * lane, lds_L + 1024u * (unsigned)p);
    }
}

// sum of v over the wave on the DPP network (row shifts, then the two row broadcasts): the total lands in lane 63
__device__ __forceinline__ float decim_wave_total_lane63(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true)); // row_shr:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true)); // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true)); // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true)); // row_shr:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true)); // row_bcast:15 into rows 1, 3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, true)); // row_bcast:31 into rows 2, 3
    return v;
}

template <bool IIR>
__global__ __launch_bounds__(kDfT, 4) void fir_decim_fd_kernel(DecimFdArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem_df[]; // the ONLY LDS object
    const int t0   = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t0 >> 6), lane0 = t0 & 63;
    // ---- kernel-lifetime registers
    float2 twX[8], tw1[8], tw2[8], Rr[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        twX[k] = a.twX[t0 * 8 + k];
        tw1[k] = a.tw1[lane0 * 8 + k];
        tw2[k] = a.tw2[(lane0 & 7) * 8 + k];
        Rr[k]  = a.R[(wave * 64 + lane0) * 8 + k];
    }
    const float2 bI1 = a.twI1[lane0 & 15], bI2 = a.twI2[t0 & 127];
    const unsigned lds_L = __builtin_amdgcn_readfirstlane(lds_off(smem_df));
    long blk = blockIdx.x, blk_end = a.n_blocks, emit0 = 0;
    // IIR: a contiguous run of blocks per workgroup (the cascade's state is carried from block to block in the cascade wave's registers); every run but the
    // span's first starts warm_blocks early from zero state, those blocks advance the state only
    float  yc[14];  // the cascade wave's samples of the block before this one
    float* Yl = reinterpret_cast<float*>(smem_df + kDfLdsAll);
    float* Tl = Yl + 896; // [4][2] the cascade's carried state, per section (LDS: touched by the cascade wave only)
    long   pblk = -1; // block whose decimated samples sit in Yl
    if constexpr (IIR) {
        emit0 = (long)blockIdx.x * a.run_blocks;
        if (emit0 >= a.n_blocks) return;
        blk     = emit0 - a.warm_blocks < 0 ? 0 : emit0 - a.warm_blocks;
        blk_end = emit0 + a.run_blocks < a.n_blocks ? emit0 + a.run_blocks : a.n_blocks;
        if (wave == 7 && lane0 < 8) Tl[lane0] = blk == 0 ? a.iir_state[lane0] : 0.f;
    }
    const decim_ctab_t itab = (decim_ctab_t)(uintptr_t)a.iir;
    // the cascade on the block in Yl: sections [s_lo, s_hi) (wave 7; between two of the block loop's barriers each)
    const auto cascade = [&](int s_lo, int s_hi, int lane) {
#ifdef GR4_DFIIR_NOCASCADE // (timing only: what the fused launch costs without the cascade's arithmetic)
        return;
#endif
#pragma unroll 1
        for (int s = s_lo; s < s_hi && s < a.iir_nsec; ++s) decim_iir_section(yc, itab + 32 * s, Tl + 2 * s, lane);
    };
    const auto cascade_load = [&](int lane) {
#pragma unroll
        for (int i = 0; i < 7; ++i) { const float2 v = *reinterpret_cast<const float2*>(Yl + 14 * lane + 2 * i); yc[2 * i] = v.x; yc[2 * i + 1] = v.y; }
    };
    const auto cascade_store = [&](int lane) {
        if (pblk >= emit0) {
            float* yo = a.y + pblk * (kDfHop / 8) + 14 * lane;
#pragma unroll
            for (int i = 0; i < 7; ++i) *reinterpret_cast<float2*>(yo + 2 * i) = make_float2(yc[2 * i], yc[2 * i + 1]);
        }
    };
    float pw_in = 0.f, pw_out = 0.f;
    // every block's own verdict: the workgroup's sum of out - thr * in.  A lane keeps the block's share until the next block's top barrier, the waves' totals (DPP network)
    // meet in 16 words of LDS behind the image, wave 0 collects the sum of the block before -- no barrier of its own
    float  pw_dprev = 0.f, pw_dmin = 0.f;
    float* Gv = reinterpret_cast<float*>(smem_df + kDfLds);
    if (threadIdx.x < 16) Gv[threadIdx.x] = 0.f;
    int   iter  = 0;
    if (blk < a.n_blocks) decim_dma(a, blk, lds_L, wave, lane0);
    for (; blk < blk_end; blk += IIR ? 1 : gridDim.x, ++iter) {
        const bool measure = a.pw != nullptr && blk != a.tail_blk && (!IIR || blk >= emit0); // wave-uniform; EVERY block (once): one the guard does not look at is one it cannot vouch for
        float      blk_in  = 0.f;
        int t = threadIdx.x;
        asm volatile("" : "+v"(t));
        const int l = t & 63;
        float2*   R = reinterpret_cast<float2*>(smem_df + kDfOffW + wave * kDfRS);
        G16_FULL_BARRIER(); // T: the block has landed; everybody is done with the previous one
        if (a.pw != nullptr) {
            const float wt = decim_wave_total_lane63(pw_dprev); // the block before this one
            if ((threadIdx.x & 63) == 63) Gv[8 * (iter & 1) + wave] = wt;
            if (wave == 0) {
                const float4 g0 = *reinterpret_cast<const float4*>(Gv + 8 * ((iter + 1) & 1)), g1 = *reinterpret_cast<const float4*>(Gv + 8 * ((iter + 1) & 1) + 4);
                pw_dmin = fminf(pw_dmin, ((g0.x + g0.y) + (g0.z + g0.w)) + ((g1.x + g1.y) + (g1.z + g1.w)));
            }
            pw_dprev = 0.f;
        }
        // ---- phase 0: radix-8 across the block's eight 512-point segments
        float2 u[8];
        {
            f2v            d[8];
            const unsigned la = lds_L + 8u * (unsigned)t;
            G16_RD8(d, la, 4096);
            lds_wait8(d);
            unpack8(u, d);
        }
        if (measure) {
#pragma unroll
            for (int q = 0; q < 8; ++q) blk_in = fmaf(u[q].x, u[q].x, fmaf(u[q].y, u[q].y, blk_in));
            pw_in += blk_in;
        }
        fft8(u);
#pragma unroll
        for (int q = 1; q < 8; ++q) u[q] = cmul(u[q], twX[q]);
#pragma unroll
        for (int q = 0; q < 8; ++q) *reinterpret_cast<float2*>(smem_df + kDfOffW + q * kDfRS + 8 * t) = u[q];
        G16_LDS_BARRIER(); // #1: the landing buffer is free
        const long bn = IIR ? (blk + 1 < blk_end ? blk + 1 : blk) : (blk + gridDim.x < a.n_blocks) ? blk + gridDim.x : blk; // (last iteration re-reads its own block: no divergent paths around the DMA)
        decim_dma(a, bn, lds_L, wave, l);
        // ---- phase 1: wave q: Z[8 k' + q], 512 points on its own
        float2 v[8];
        {
            f2v            d[8];
            const unsigned a0 = lds_off(R) + 8u * (unsigned)l;
            G16_RD8(d, a0, 512);
            lds_wait8(d);
            unpack8(v, d);
        }
        fft8(v);
#pragma unroll
        for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], tw1[k]);
        private_tail(v, R, l, tw2);
        // ---- G[r] = sum_j R[r + 1024 j] Z[r + 1024 j]: r = 8 k'_r + q, k'_r = bin(lane, kb1 in {0, 1}); + 1024 j <-> register kb1 + 2 j
        float2 g0 = cmul(Rr[0], v[0]), g1 = cmul(Rr[1], v[1]);
#pragma unroll
        for (int j = 1; j < 4; ++j) {
            g0 = cadd(g0, cmul(Rr[2 * j], v[2 * j]));
            g1 = cadd(g1, cmul(Rr[2 * j + 1], v[2 * j + 1]));
        }
        // ---- G_q[k'], k' = (l >> 3) + 8 (l & 7) + 64 kb1, into row q of the shared G / F buffer
        float2* GF = reinterpret_cast<float2*>(smem_df + kDfOffF);
        {
            const int kk = 129 * wave + (l >> 3) + 8 * (l & 7);
            GF[kk]      = g0;
            GF[kk + 64] = g1;
        }
        G16_LDS_BARRIER(); // #2: all eight G_q are in place
        if constexpr (IIR) { // wave 7 is idle from here to the top of the loop (inverse transforms: waves 0, 1; final pass: lanes 0 .. 127): the block before this one
#ifdef GR4_DFIIR_PIPE_TIMING // (timing only, results are garbage: three waves, GR4_DFIIR_PIPE_TIMING sections each per block -- is there room for a pipeline of waves?)
            if (wave >= 5 && pblk >= 0) {
                cascade_load(l);
                cascade(0, 1, l);
            }
#else
            if (wave == 7 && pblk >= 0) { // goes through the cascade meanwhile, half of it before barrier #3 ...
                cascade_load(l);
                cascade(0, 2, l);
            }
#endif
        }
        // ---- the eight 128-point inverse transforms F_q[i''] = sum_k' G_q[k'] W_128^{-k' i''}: waves 0 and 1, one transform per 16-lane group (8 points a lane),
        //      so that every lane of the two waves works (one transform per wave on 16 of its 64 lanes cost four times the issue slots)
        if (wave < 2) {
            const int      grp = l >> 4, kl = l & 15;            // transform q = 4 wave + grp, lane kl of its sixteen
            float2*        Gq  = GF + 129 * (4 * wave + grp);
            float2*        X   = R + 136 * grp;                  // exchange scratch of this group inside the wave's (now idle) private region: 8 rows of 17
            f2v            d[8];
            const unsigned a1 = lds_off(Gq) + 8u * (unsigned)kl;
            G16_RD8(d, a1, 128); // G_q[kl + 16 k_hi]
            lds_wait8(d);
            unpack8(v, d);
            ifft8(v);            // -> a1 = 0..7
            mul_powers(v, bI1);  // W_128^{-k_lo a1}
            // exchange inside the 16 lanes: lane (a1', hh) takes k_lo = j + 8 hh, j = 0..7
#pragma unroll
            for (int k = 0; k < 8; ++k) X[17 * k + kl] = v[k]; // [a1][k_lo], pitch 17
            const int      a1p = kl & 7, hh = kl >> 3;
            const unsigned a2  = lds_off(X) + 8u * (unsigned)(17 * a1p + 8 * hh);
            G16_RD8(d, a2, 8);
            lds_wait8(d);
            unpack8(v, d);
            // radix-2 over hh between lanes l and l ^ 8 (row rotate by 8), then W_16^{-j} on the difference lanes, then radix-8 over j
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float px = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[j].x), 0x128, 0xF, 0xF, false)); // row_ror:8
                const float py = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[j].y), 0x128, 0xF, 0xF, false));
                v[j] = hh ? make_float2(px - v[j].x, py - v[j].y) : make_float2(v[j].x + px, v[j].y + py); // hh = 0: A[j] + A[j + 8];  hh = 1: A[j] - A[j + 8]
            }
            if (hh) { // times W_16^{-j} = conj(W_16^j)
                constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, hq = 0.70710678118654752440f;
                v[1] = cmul(v[1], make_float2(c1, s1));
                v[2] = make_float2((v[2].x - v[2].y) * hq, (v[2].y + v[2].x) * hq);
                v[3] = cmul(v[3], make_float2(s1, c1));
                v[4] = make_float2(-v[4].y, v[4].x);
                v[5] = cmul(v[5], make_float2(-s1, c1));
                v[6] = make_float2((-v[6].x - v[6].y) * hq, (v[6].x - v[6].y) * hq);
                v[7] = cmul(v[7], make_float2(-c1, s1));
            }
            ifft8(v); // -> b' : i'' = a1' + 8 (2 b' + hh); F_q overwrites G_q (every lane of the group has read its inputs: one wave, in order)
#pragma unroll
            for (int bp = 0; bp < 8; ++bp) Gq[a1p + 8 * (2 * bp + hh)] = v[bp];
        }
        G16_LDS_BARRIER(); // #3: every F_q is in place (IIR: and wave 7 has taken the previous block out of Yl)
        if constexpr (IIR) {
#ifdef GR4_DFIIR_PIPE_TIMING
            if (wave >= 5 && pblk >= 0) {
                if (GR4_DFIIR_PIPE_TIMING > 1) cascade(1, 2, l);
                if (wave == 7) cascade_store(l);
            }
#else
            if (wave == 7 && pblk >= 0) { // ... and half of it behind
                cascade(2, 4, l);
                cascade_store(l);
            }
#endif
        }
        // ---- final pass across the waves: y[i'' + 128 a] = Re( sum_q W_1024^{-q i''} F_q[i''] W_8^{-q a} ), lanes 0..127
        if (t < 128) {
            const float2* F = reinterpret_cast<const float2*>(smem_df + kDfOffF) + t;
#pragma unroll
            for (int q = 0; q < 8; ++q) v[q] = F[129 * q];
            mul_powers(v, bI2); // W_1024^{-q i''}
            ifft8(v);
            // valid outputs: i' = i'' + 128 a >= 128, i.e. a = 1..7 -> y[blk * 896 + i' - 128]
            float*    yo    = a.y + blk * (kDfHop / 8) + t;
            const int valid = blk == a.tail_blk ? a.tail_out : kDfHop / 8;
            if constexpr (IIR) { // the decimated samples stay on chip: to the cascade wave through LDS (read after the next barrier)
#pragma unroll
                for (int aa = 1; aa < 8; ++aa) Yl[t + 128 * (aa - 1)] = v[aa].x;
            } else {
#pragma unroll
                for (int aa = 1; aa < 8; ++aa)
                    if (t + 128 * (aa - 1) < valid) yo[128 * (aa - 1)] = v[aa].x;
            }
            if (measure) {
                float blk_out = 0.f;
#pragma unroll
                for (int aa = 1; aa < 8; ++aa) blk_out = fmaf(v[aa].x, v[aa].x, blk_out);
                pw_out += blk_out;
                pw_dprev = fmaf(-a.pw_thr, blk_in, blk_out);
            }
        }
        if constexpr (IIR) {
            pblk = blk;
            // the cross-pass twiddles are needed again only behind the next top barrier: fetched anew here (L2), so that their 16 registers are free while the
            // cascade wave holds its 14 samples (laundered pointer: the loads must not be hoisted back out of the loop)
            const float2* p = a.twX + t0 * 8;
            asm volatile("" : "+v"(p));
#pragma unroll
            for (int k = 0; k < 8; ++k) twX[k] = p[k];
        }
    }
    if constexpr (IIR) { // the run's last block: its samples are in Yl behind the final pass
        G16_LDS_BARRIER();
        if (wave == 7 && pblk >= 0) {
            const int l = threadIdx.x & 63;
            cascade_load(l);
            cascade(0, 4, l);
            cascade_store(l);
            if (blk_end == a.n_blocks && l < 8) a.iir_state_out[l] = Tl[l]; // the state behind the span
        }
    }
    if (a.pw != nullptr) { // one pair of atomics per wave, spread over 16 slots
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            pw_in += __shfl_xor(pw_in, off);
            pw_out += __shfl_xor(pw_out, off);
        }
        {   // the last two blocks' verdicts
            const float wt = decim_wave_total_lane63(pw_dprev);
            if ((threadIdx.x & 63) == 63) Gv[8 * (iter & 1) + wave] = wt;
        }
        // one pair of atomics per WORKGROUP, spread over 16 slots (a pair per wave -- 8192 atomics on 32 addresses -- cost ~35 us at the end of a 176 us launch)
        __syncthreads(); // every lane is past the final pass: the G / F buffer is free (the landing buffer is not: the last block's re-read may still be in flight)
        float* red = reinterpret_cast<float*>(smem_df + kDfOffF);
        if ((threadIdx.x & 63) == 0) { red[2 * (threadIdx.x >> 6)] = pw_in; red[2 * (threadIdx.x >> 6) + 1] = pw_out; }
        __syncthreads();
        if (threadIdx.x == 0) { // the last workgroup to finish folds the slots, re-arms them and hands the totals to the host (chain_fused.hip: same scheme)
            float si = 0.f, so = 0.f, s0 = 0.f, s1 = 0.f;
#pragma unroll
            for (int w = 0; w < kDfT / 64; ++w) { si += red[2 * w]; so += red[2 * w + 1]; s0 += Gv[w]; s1 += Gv[8 + w]; }
            if (fminf(pw_dmin, fminf(s0, s1)) < 0.f) atomicOr(reinterpret_cast<unsigned*>(a.pw + 33), 1u);
            float* slot = a.pw + 2 * (blockIdx.x & 15);
            atomicAdd(slot, si);
            atomicAdd(slot + 1, so);
            __threadfence();
            unsigned* done = reinterpret_cast<unsigned*>(a.pw + 32);
            if (atomicAdd(done, 1u) == gridDim.x - 1) {
                __threadfence();
                float tin = 0.f, tout = 0.f;
                for (int k = 0; k < 16; ++k) { tin += atomicExch(a.pw + 2 * k, 0.f); tout += atomicExch(a.pw + 2 * k + 1, 0.f); }
                atomicExch(done, 0u);
                reinterpret_cast<volatile unsigned*>(a.pw_host)[3] = atomicExch(reinterpret_cast<unsigned*>(a.pw + 33), 0u); // a block of this launch fell below the threshold
                *reinterpret_cast<volatile unsigned long long*>(a.pw_host) = (unsigned long long)__float_as_uint(tin) | ((unsigned long long)__float_as_uint(tout) << 32);
                __threadfence_system();
                reinterpret_cast<volatile unsigned*>(a.pw_host)[2] = a.pw_seq;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct DecimFdIir { const float* tab; int nsec; const float* state_in; float* state_out; int warm_blocks; }; // (device pointers; fir.hip fills it from the IIR handle)
struct FirDecimFd {
    DeviceBuffer d_twX, d_tw1, d_tw2, d_R, d_twI1, d_twI2, d_hist1024, d_stage, d_pw;
    float*       h_pw = nullptr;     // page-locked, device-mapped {in, out, sequence number} of the last measured launch that has finished
    float*       d_hpw = nullptr;    // its device view
    unsigned     pw_seq = 0, pw_seen = 0;
    hipStream_t  pw_stream = nullptr;
    ~FirDecimFd() {
        if (h_pw) hip_quiet(hipHostFree(h_pw));
    }
};

static void put_wd(std::vector<float>& v, size_t idx, long num, long den) { // exp(-2 pi i num / den)
    const double ang = -2.0 * M_PI * (double)(((num % den) + den) % den) / (double)den;
    v[2 * idx]     = (float)std::cos(ang);
    v[2 * idx + 1] = (float)std::sin(ang);
}
template <typename T>
static int upload_df(DeviceBuffer& b, const std::vector<T>& h) {
    int rc = b.ensure(h.size() * sizeof(T));
    if (rc) return rc;
    GR4_HIP_TRY(upload_fresh(b.ptr, h.data(), h.size() * sizeof(T)));
    return GR4HIP_OK;
}

int fir_decim_fd_supported(size_t ntaps, size_t decim) { return decim == 8 && ntaps >= 1 && ntaps <= (size_t)kDfV + 1; }

int fir_decim_fd_create(FirDecimFd** out, const float* taps, size_t ntaps) {
    auto* c = new (std::nothrow) FirDecimFd();
    GR4_REQUIRE(c, "out of host memory");
    using cd = std::complex<double>;
    const int NC = kDfN / 2;
    std::vector<cd> H(NC + 1), Rt(NC);
    for (int k = 0; k <= NC; ++k) { // H[k] = sum_j b[j] e^{-2 pi i j k / 8192}, float64, exact angle reduction
        cd s = 0;
        for (size_t j = 0; j < ntaps; ++j) {
            const double ang = -2.0 * M_PI * (double)((j * (size_t)k) % kDfN) / kDfN;
            s += (double)taps[j] * cd(std::cos(ang), std::sin(ang));
        }
        H[k] = s;
    }
    auto W = [](int k) { const double ang = -2.0 * M_PI * (double)k / kDfN; return cd(std::cos(ang), std::sin(ang)); };
    const cd I(0, 1);
    auto P = [&](int k) { return H[k] * (1.0 - I * W(k)) * 0.5; };
    auto Q = [&](int k) { return H[k] * (1.0 + I * W(k)) * 0.5; };
    for (int k = 0; k < NC; ++k) Rt[k] = P(k) + std::conj(Q(NC - k));
    Rt[0] = (Rt[0] + (P(NC) + std::conj(Q(0)))) * 0.5;
    const double norm = 1.0 / (4.0 * 1024.0); // Re(.) / 4 and the 1 / 1024 of the inverse transform
    std::vector<float> twX(2 * 512 * 8), tw1(2 * 64 * 8), tw2(2 * 8 * 8), Rq(2 * 8 * 64 * 8), tI1(2 * 16), tI2(2 * 128);
    for (int m = 0; m < 512; ++m)
        for (int q = 0; q < 8; ++q) put_wd(twX, (size_t)m * 8 + q, (long)m * q, 4096);
    for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 8; ++k) put_wd(tw1, (size_t)l * 8 + k, (long)l * k, 512);
    for (int l0 = 0; l0 < 8; ++l0)
        for (int k = 0; k < 8; ++k) put_wd(tw2, (size_t)l0 * 8 + k, (long)l0 * k, 64);
    for (int q = 0; q < 8; ++q)
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 8; ++r) {
                const cd     val = Rt[8 * c16_out_bin(l, r) + q] * norm;
                const size_t i   = ((size_t)q * 64 + l) * 8 + r;
                Rq[2 * i]     = (float)val.real();
                Rq[2 * i + 1] = (float)val.imag();
            }
    for (int kl = 0; kl < 16; ++kl) put_wd(tI1, (size_t)kl, -(long)kl, 128);
    for (int i = 0; i < 128; ++i) put_wd(tI2, (size_t)i, -(long)i, 1024);
    int rc = upload_df(c->d_twX, twX);
    if (!rc) rc = upload_df(c->d_tw1, tw1);
    if (!rc) rc = upload_df(c->d_tw2, tw2);
    if (!rc) rc = upload_df(c->d_R, Rq);
    if (!rc) rc = upload_df(c->d_twI1, tI1);
    if (!rc) rc = upload_df(c->d_twI2, tI2);
    if (rc) { delete c; return rc; }
    *out = c;
    return GR4HIP_OK;
}
void fir_decim_fd_destroy(FirDecimFd* c) { delete c; }

// n_blocks blocks of 7168 input samples -> 896 outputs each; d_hist1024: the 1024 samples in front of d_in
// ONE small launch in front of the transform kernel: the block's history widened to the 1024 samples the first block reads, and (tail_n > 0) the staging image of
// the span's last, partial block: stream positions full_blocks * 7168 - 1024 .. + 8191, zeros from n_in on
__global__ __launch_bounds__(256) void decim_fd_prepare_kernel(const float* __restrict__ hist, int hcap, const float* __restrict__ x, long n_in, long full_blocks, float* __restrict__ hist1024,
                                                               float* __restrict__ stage, int with_tail) {
    const int  i   = blockIdx.x * 256 + threadIdx.x; // 0 .. 1023: history; 1024 .. 9215: staging image
    const auto at  = [&](long pos) -> float { return pos >= n_in ? 0.f : pos >= 0 ? x[pos] : pos >= -(long)hcap ? hist[hcap + pos] : 0.f; };
    if (i < kDfV) hist1024[i] = at((long)i - kDfV);
    else if (with_tail && i < kDfV + kDfN) stage[i - kDfV] = at(full_blocks * kDfHop - kDfV + (i - kDfV));
}

// d_hist: the filter's history (hcap samples in front of d_in); n_in input samples (a multiple of 8): floor(n_in / 7168) whole blocks and, if anything is left, one partial block
// measure: this launch samples input and output power (dynamic-range guard, fir.hip); fir_decim_fd_power_ratio hands the result out
// iir != null: whole blocks only (n_in a multiple of 7168), the biquad cascade rides in the launch and d_out receives ITS output (fir_decim_fd_kernel<true>)
int fir_decim_fd_run(FirDecimFd* c, const float* d_in, size_t n_in, const float* d_hist, int hcap, float* d_out, hipStream_t st, bool measure, const DecimFdIir* iir) {
    const size_t full = n_in / kDfHop, rest = n_in - full * kDfHop;
    GR4_REQUIRE(!iir || rest == 0, "fir_decim_fd: the fused cascade takes whole blocks");
    int          rc   = c->d_hist1024.ensure(kDfV * sizeof(float));
    if (!rc && rest) rc = c->d_stage.ensure(kDfN * sizeof(float));
    if (rc) return rc;
    hipLaunchKernelGGL(decim_fd_prepare_kernel, dim3((kDfV + (rest ? kDfN : 0)) / 256), dim3(256), 0, st, d_hist, hcap, d_in, (long)n_in, (long)full, (float*)c->d_hist1024.ptr,
                       (float*)c->d_stage.ptr, rest ? 1 : 0);
    GR4_LAUNCH_CHECK();
    const size_t n_blocks = full + (rest ? 1 : 0);
    DecimFdArgs a{};
    a.x = d_in; a.hist = (const float*)c->d_hist1024.ptr;
    a.x_tail = (const float*)c->d_stage.ptr; a.tail_blk = rest ? (long)full : -1; a.tail_out = (int)(rest / 8);
    a.twX = static_cast<const float2*>(c->d_twX.ptr); a.tw1 = static_cast<const float2*>(c->d_tw1.ptr); a.tw2 = static_cast<const float2*>(c->d_tw2.ptr);
    a.R = static_cast<const float2*>(c->d_R.ptr); a.twI1 = static_cast<const float2*>(c->d_twI1.ptr); a.twI2 = static_cast<const float2*>(c->d_twI2.ptr);
    a.y = d_out; a.n_blocks = (long)n_blocks;
    if (measure) {
        if (!c->h_pw) {
            rc = c->d_pw.ensure(36 * sizeof(float));
            if (rc) return rc;
            GR4_HIP_TRY(hipMemsetAsync(c->d_pw.ptr, 0, 36 * sizeof(float), st)); // (in front of the first measured launch, on its stream)
            GR4_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_pw), 4 * sizeof(float), hipHostMallocMapped));
            std::memset(c->h_pw, 0, 4 * sizeof(float));
            GR4_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_hpw), c->h_pw, 0));
        }
        a.pw        = static_cast<float*>(c->d_pw.ptr);
        a.pw_host   = c->d_hpw;
        a.pw_seq    = ++c->pw_seq;
        a.pw_thr    = kDecimFdBlockThreshold * (float)(kDfHop / 8) / (float)kDfN; // (the scale fir_decim_fd_power_ratio takes out)
        c->pw_stream = st;
    }
    static PerDevice per_device;
    bool             first = false;
    int              dev = -1, n_cu = per_device.current(&first, &dev);
    GR4_REQUIRE(n_cu != 0, "fir_decim_fd: cannot query the current device");
    if (first) {
        n_cu = -n_cu;
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fir_decim_fd_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, kDfLdsAll));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fir_decim_fd_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, kDfLdsIir));
        per_device.done(dev, n_cu);
    }
    if (iir) { // contiguous runs: as many as there are workgroup slots, none shorter than its own warm-up
        const size_t slots = (size_t)2 * n_cu;
        const size_t run   = std::max<size_t>(ceil_div(n_blocks, slots), (size_t)std::max(iir->warm_blocks, 1));
        a.iir = iir->tab; a.iir_nsec = iir->nsec; a.iir_state = iir->state_in; a.iir_state_out = iir->state_out;
        a.run_blocks = (int)run; a.warm_blocks = iir->warm_blocks;
        hipLaunchKernelGGL(fir_decim_fd_kernel<true>, dim3((unsigned)ceil_div(n_blocks, run)), dim3(kDfT), kDfLdsIir, st, a);
        GR4_LAUNCH_CHECK();
        return GR4HIP_OK;
    }
    const unsigned grid = (unsigned)std::min<size_t>(n_blocks, (size_t)2 * n_cu);
    hipLaunchKernelGGL(fir_decim_fd_kernel<false>, dim3(grid), dim3(kDfT), kDfLdsAll, st, a);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}
// mean output power / mean input power of the last measured launch's sampled blocks.  wait: synchronise on it; otherwise only when it has finished.
// Returns 1 with *ratio set, 0 if nothing (new) is available.
int fir_decim_fd_power_ratio(FirDecimFd* c, bool wait, float* ratio) {
    if (!c->h_pw || c->pw_seq == c->pw_seen) return 0;
    volatile unsigned* seqw = reinterpret_cast<volatile unsigned*>(c->h_pw) + 2;
    if (wait) { // spin on the mapped word (a few microseconds behind the last workgroup); a stream that has finished without writing it ends the wait
        for (unsigned long spins = 1; *seqw != c->pw_seq; ++spins) {
            if ((spins & 4095) == 0 && hipStreamQuery(c->pw_stream) != hipErrorNotReady) {
                if (hipStreamSynchronize(c->pw_stream) != hipSuccess || *seqw != c->pw_seq) return 0;
                break;
            }
            __builtin_ia32_pause();
        }
    } else if (*seqw == c->pw_seen) {
        return 0;
    }
    const unsigned long long word = *reinterpret_cast<volatile unsigned long long*>(c->h_pw);
    c->pw_seen = *seqw;
    float pair[2];
    std::memcpy(pair, &word, sizeof(pair));
    const double in = pair[0], out = pair[1];
    // per sampled block: 8192 input samples (the overlap counted twice: statistics only), 896 outputs
    *ratio = in > 0 ? (float)((out / (kDfHop / 8)) / (in / kDfN)) : 1.f;
    if (reinterpret_cast<volatile unsigned*>(c->h_pw)[3] != 0u && *ratio >= kDecimFdBlockThreshold) *ratio = 0.5f * kDecimFdBlockThreshold; // one block below the threshold is enough
    return 1;
}

} // namespace gr4
