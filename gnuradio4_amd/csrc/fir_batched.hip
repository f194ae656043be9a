// fir_batched.hip -- many-channel FIR as a block-Toeplitz contraction on the f32 MFMA units (BASELINE.json configs[3]).
//
// nchannels independent gr::filter::fir_filter<float> instances (blocks/filter/.../time_domain_filter.hpp:22-48), per-channel taps.
// With Kp = ntaps rounded up to 16 and the output index split as n = 16 i + j:
//     y_c[16 i + j] = sum_{u=0}^{Kp+15} A_c[j][u] * B_c[u][i],   A_c[j][u] = b_c[Kp + j - u],   B_c[u][i] = x_c[16 i - Kp + u]
// i.e. per channel a [16 x (Kp+16)] x [(Kp+16) x Nblocks] product whose B operand is just a sliding window of the input: a real dense
// contraction with (Kp+16)/Kp = 6 % padding waste (instead of the 2x of a naive Toeplitz GEMM).  v_mfma_f32_16x16x4_f32 has no rate
// advantage over the FP32 VALU on gfx950 (both 64 flop/clk/SIMD, MI355X_MICROARCH.md) but reaches that rate from one wave per SIMD
// with one VGPR per operand: the A fragments (taps) of a channel live in (Kp+16)/4 registers for the whole workgroup, the B operand
// is one conflict-free ds_read_b32 per MFMA from the staged (18/16-padded) input segment, D leaves as fully coalesced float4 stores.
// Bound: MFMA f32 (157.3 TFLOP/s): 2*(Kp+16) flop per output sample.
#include "common.hpp"
#include "fir_exact.hpp"
#include "buffer_ops.hpp"

#include <cstdlib>

namespace gr4 {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kSeg = 4096;    // output samples per segment (4 waves x 4 tiles x 256)
constexpr int kSegPerWg = 4;  // consecutive segments per workgroup

template <int KS> // K-steps of 4: Kp = 4 KS - 16
__global__ __launch_bounds__(256) void fir_mfma_kernel(const float* __restrict__ x, long in_stride, const float* __restrict__ hist, const float* __restrict__ afrag,
                                                        float* __restrict__ y, long out_stride, long n, float* __restrict__ new_hist /*single stream: the next Kp-sample history, or null*/) {
    constexpr int Kp   = 4 * KS - 16;
    constexpr int NPAD = (kSeg + Kp) / 16 * 18; // two pad floats per 16 samples: block stride 18 = 2 mod 32 banks, so the 32 lanes (16 blocks x 2 K
                                                // offsets) of a ds_read_b32 group hit 32 different banks (stride 17 puts two of them on one)
    constexpr int NL   = (kSeg + Kp + 255) / 256; // samples a lane holds for the next segment
    __shared__ float xs[NPAD];
    const int  c    = blockIdx.y;
    const int  tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xc = x + (long)c * in_stride;
    const float* hc = hist + (long)c * Kp;
    float*       yc = y + (long)c * out_stride;

    float a[KS]; // A fragments: lane l holds A[j = l & 15][u = 4 ks + (l >> 4)] = b[Kp + j - u]
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = afrag[((long)c * KS + ks) * 64 + lane];

    // A workgroup takes kSegPerWg consecutive segments of its channel; the samples of segment s + 1 are requested (into registers) before the
    // MFMAs of segment s and written to LDS after them.  The workgroups of a launch start together and stay in step: without the prefetch the
    // chip alternates between everybody waiting for HBM and everybody on the matrix pipe (the same fix took fir_mfma_decim_kernel from 0.55 to 0.39 ms).
    float nxt[NL];
    auto load_next = [&](long seg0) { // seg0 >= kSeg > Kp: no index below 0; past the end of the span / of the segment the range check returns 0
        const long   i0   = seg0 - Kp;
        const long   nrec = n - i0 < (long)(kSeg + Kp) ? n - i0 : (long)(kSeg + Kp);
        const rsrc_t r    = make_rsrc(xc + i0, (unsigned)(nrec > 0 ? nrec * 4 : 0));
#pragma unroll
        for (int u = 0; u < NL; ++u) nxt[u] = buf_load_f(r, tid * 4, 256 * u * 4);
    };
    const long nseg = (n + kSeg - 1) / kSeg, sfirst = (long)blockIdx.x * kSegPerWg, slast = sfirst + kSegPerWg < nseg ? sfirst + kSegPerWg : nseg;
    if (sfirst > 0 && sfirst < slast) load_next(sfirst * kSeg);
    const int col = lane & 15, kq = lane >> 4;
    for (long sg = sfirst; sg < slast; ++sg) {
        const long seg0 = sg * kSeg;
        if (sg > 0) {
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int s_ = tid + 256 * u;
                if (s_ < kSeg + Kp) xs[s_ + 2 * (s_ >> 4)] = nxt[u];
            }
        } else {
            for (int s_ = tid; s_ < kSeg + Kp; s_ += 256) { // the first segment of the span reads the carried history in front of x
                const long i = seg0 - Kp + s_;
                xs[s_ + 2 * (s_ >> 4)] = i >= 0 ? (i < n ? xc[i] : 0.f) : hc[Kp + i];
            }
        }
        __syncthreads();
        if (sg + 1 < slast) load_next(seg0 + kSeg); // in flight during the MFMAs below
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) { // two independent accumulators hide the 40-cycle dependent MFMA latency
            const int ib0 = 16 * (4 * wave + 2 * pair), ib1 = ib0 + 16; // first 16-sample block of each tile
            f32x4     acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
            const float* p0 = xs + 18 * (ib0 + col) + kq;
            const float* p1 = xs + 18 * (ib1 + col) + kq;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = 4 * ks + 2 * (ks >> 2); // padded offset of u = 4 ks within the window
                acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], p0[off], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], p1[off], acc1, 0, 0, 0);
            }
            // D[row = 4 kq + r][col]: y[16 (ib + col) + 4 kq + r]
            const long o0 = seg0 + 16L * (ib0 + col) + 4 * kq, o1 = seg0 + 16L * (ib1 + col) + 4 * kq;
            if (o0 + 3 < n) *reinterpret_cast<float4*>(yc + o0) = make_float4(acc0[0], acc0[1], acc0[2], acc0[3]);
            else
                for (int r = 0; r < 4; ++r)
                    if (o0 + r < n) yc[o0 + r] = acc0[r];
            if (o1 + 3 < n) *reinterpret_cast<float4*>(yc + o1) = make_float4(acc1[0], acc1[1], acc1[2], acc1[3]);
            else
                for (int r = 0; r < 4; ++r)
                    if (o1 + r < n) yc[o1 + r] = acc1[r];
        }
        __syncthreads(); // every wave is done with the staged segment before the next one overwrites it
    }
    if (new_hist != nullptr && blockIdx.x == 0 && blockIdx.y == 0) { // (the other half of the caller's ping-pong pair: nobody reads it in this launch)
        for (int h = tid; h < Kp; h += 256) {
            const long i = n - Kp + h;
            new_hist[h]  = i >= 0 ? x[i] : hist[Kp + i];
        }
    }
}

// fir_filter<complex<float>> (real taps on interleaved {re, im} samples) on the same scheme: the two components are two real streams under the same
// taps.  The segment is de-interleaved into a re plane and an im plane while it is staged; a wave's two accumulators are the re and the im tile of the
// same 256 outputs (same A fragment, B operand from the two planes), so D leaves re-interleaved as two 16-byte stores per lane.  2048 complex outputs
// per segment (the same 4096 real outputs and the same LDS footprint as fir_mfma_kernel).  hist: the Kp complex samples in front of x.
constexpr int kSegC = 2048;
template <int KS>
__global__ __launch_bounds__(256) void fir_mfma_c32_kernel(const float2* __restrict__ x, const float2* __restrict__ hist, const float* __restrict__ afrag, float2* __restrict__ y, long n,
                                                            float2* __restrict__ new_hist /*the next Kp-sample history, or null*/) {
    constexpr int Kp   = 4 * KS - 16;
    constexpr int NPAD = (kSegC + Kp) / 16 * 18 + 16; // per plane; + 16: the two planes sit 16 banks apart, so the re / im halves of a staged lane pair do not collide
    constexpr int NL   = (kSegC + Kp + 255) / 256;    // complex samples a lane holds for the next segment
    __shared__ float xs[2 * NPAD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    float     a[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = afrag[ks * 64 + lane];
    float2 nxt[NL];
    auto   load_next = [&](long seg0) {
        const long   i0   = seg0 - Kp;
        const long   nrec = n - i0 < (long)(kSegC + Kp) ? n - i0 : (long)(kSegC + Kp);
        const rsrc_t r    = make_rsrc(x + i0, (unsigned)(nrec > 0 ? nrec * 8 : 0));
#pragma unroll
        for (int u = 0; u < NL; ++u) nxt[u] = buf_load_f2(r, tid * 8, 256 * u * 8);
    };
    const long nseg = (n + kSegC - 1) / kSegC, sfirst = (long)blockIdx.x * kSegPerWg, slast = sfirst + kSegPerWg < nseg ? sfirst + kSegPerWg : nseg;
    if (sfirst > 0 && sfirst < slast) load_next(sfirst * kSegC);
    const int col = lane & 15, kq = lane >> 4;
    for (long sg = sfirst; sg < slast; ++sg) {
        const long seg0 = sg * kSegC;
        if (sg > 0) {
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int s_ = tid + 256 * u;
                if (s_ < kSegC + Kp) {
                    xs[s_ + 2 * (s_ >> 4)]        = nxt[u].x;
                    xs[NPAD + s_ + 2 * (s_ >> 4)] = nxt[u].y;
                }
            }
        } else {
            for (int s_ = tid; s_ < kSegC + Kp; s_ += 256) { // the first segment of the span reads the carried history in front of x
                const long   i = seg0 - Kp + s_;
                const float2 v = i >= 0 ? (i < n ? x[i] : make_float2(0.f, 0.f)) : hist[Kp + i];
                xs[s_ + 2 * (s_ >> 4)]        = v.x;
                xs[NPAD + s_ + 2 * (s_ >> 4)] = v.y;
            }
        }
        __syncthreads();
        if (sg + 1 < slast) load_next(seg0 + kSegC); // in flight during the MFMAs below
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {
            const int    ib  = 16 * (2 * wave + pair); // first 16-sample block of this wave's tile
            f32x4        acr = {0.f, 0.f, 0.f, 0.f}, aci = {0.f, 0.f, 0.f, 0.f};
            const float* pr  = xs + 18 * (ib + col) + kq;
            const float* pi  = pr + NPAD;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int off = 4 * ks + 2 * (ks >> 2);
                acr = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], pr[off], acr, 0, 0, 0);
                aci = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], pi[off], aci, 0, 0, 0);
            }
            const long o = seg0 + 16L * (ib + col) + 4 * kq; // D[row = 4 kq + r][col]: y[16 (ib + col) + 4 kq + r]
            if (o + 3 < n) {
                float4* d = reinterpret_cast<float4*>(y + o);
                d[0]      = make_float4(acr[0], aci[0], acr[1], aci[1]);
                d[1]      = make_float4(acr[2], aci[2], acr[3], aci[3]);
            } else {
                for (int r = 0; r < 4; ++r)
                    if (o + r < n) y[o + r] = make_float2(acr[r], aci[r]);
            }
        }
        __syncthreads();
    }
    if (new_hist != nullptr && blockIdx.x == 0) {
        for (int h = tid; h < Kp; h += 256) {
            const long i = n - Kp + h;
            new_hist[h]  = i >= 0 ? x[i] : hist[Kp + i];
        }
    }
}

// new_hist[c][h] = virtual input of channel c at index n - Kp + h
__global__ void fir_batched_hist_kernel(const float* __restrict__ x, long in_stride, const float* __restrict__ old_hist, float* __restrict__ new_hist, long n, int Kp) {
    const int c = blockIdx.y, h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= Kp) return;
    const long i = n - Kp + h;
    new_hist[(long)c * Kp + h] = i >= 0 ? x[(long)c * in_stride + i] : old_hist[(long)c * Kp + Kp + i];
}

// Polyphase decimating FIR on the same scheme (BASELINE configs[2]: decim 8, 1024 taps).  y[m] = sum_p sum_q b[qD + p] x_p[m - q] with
// the phase streams x_p[m] = x[mD - p]: D ordinary FIRs of Q = ceil(K / D) taps at the output rate whose products land in the same
// accumulator tile, i.e. one [16 x D (Kp+16)] x [D (Kp+16) x blocks] contraction.  The input segment is de-interleaved into D padded
// phase rows while it is staged (every input read from HBM once).  The A operand is Toeplitz -- A[j][u] = b_p[Kp + j - u] -- so it is not stored as
// fragments at all: the D phase-tap rows (Kp + 32 floats each, zero outside the taps) sit in LDS and lane (j, kq) reads its element of K-step
// ks at row[16 + Kp + j - kq - 4 ks]: one conflict-free ds_read_b32 (19 consecutive addresses per wave).  Fetching [D][KS][64] fragment tables
// from L2 instead cost every wave one 256-byte load per MFMA: the MFMA phase alone ran at 53 % of the matrix pipe.
constexpr int kDecimSPW = 4; // consecutive segments per workgroup (fir_mfma_decim_kernel)
template <int KS, int TPW> // K-steps per phase (Kp = 4 KS - 16), 256-output tiles per wave
__global__ __launch_bounds__(256) void fir_mfma_decim_kernel(const float* __restrict__ x, const float* __restrict__ hist /*[Kp D] samples in front of x*/,
                                                              const float* __restrict__ ptaps /*[D][Kp + 32]*/, float* __restrict__ y, long n_out, int D) {
    constexpr int Kp  = 4 * KS - 16;
    constexpr int TS  = Kp + 32;                     // phase-tap row: index 16 + q holds b[q D + p]
    constexpr int SEG = 1024 * TPW;                  // outputs per workgroup
    constexpr int PAD = 2;                           // floats of padding per 16 samples: stride 18 makes the B-operand reads conflict-free (as in fir_mfma_kernel)
    constexpr int ROW = (SEG + Kp) / 16 * (16 + PAD) + 1; // padded phase row (odd length: the D rows start on different banks)
    constexpr int NL  = 40;                          // samples a lane holds for the next segment: (SEG + Kp) D <= 256 NL covers D = 8 at every Kp
    extern __shared__ float xs[];                    // [D][ROW] samples, then [D][TS] taps
    float*     tp   = xs + D * ROW;
    const int  tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long n_in = n_out * D, H = (long)Kp * D;
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    const int col = lane & 15, kq = lane >> 4;
    for (int i = tid; i < D * TS; i += 256) tp[i] = ptaps[i];

    // A workgroup takes kDecimSPW consecutive segments.  The samples of segment s + 1 are requested (into registers) before the MFMA phase of
    // segment s and written to LDS after it: all workgroups of a launch start together and stay in step, so without this the chip alternates
    // between everybody waiting for HBM and everybody on the matrix pipe (measured: 0.18 ms of staging + 0.37 ms of MFMAs = the 0.55 ms total).
    // stage: phase row p, position m' <-> input index (seg0 - Kp + m') D - p.  Consecutive lanes take consecutive input samples; the
    // (m', p) pair of a lane's next sample (256 further) follows from the previous one without a division.
    const int  cnt      = (SEG + Kp) * D;
    const bool prefetch = cnt <= 256 * NL;
    const int  q256 = 256 / D, r256 = 256 % D;
    const int  e0 = tid - (D - 1);                   // = m' D - p for s = tid
    const int  mp0 = (e0 + D - 1) / D, pp0 = mp0 * D - e0;
    auto sample = [&](long seg0, int s) -> float {
        const long i = (seg0 - Kp) * D - (D - 1) + s; // lowest input index needed (m' = 0, p = D - 1) + s
        return s < cnt ? (i >= 0 ? (i < n_in ? x[i] : 0.f) : (i >= -H ? hist[H + i] : 0.f)) : 0.f;
    };
    float nxt[NL];
    auto load_next = [&](long seg0) { // seg0 >= SEG: no index is negative; SRSRC loads (one lane offset, constant per-load offsets -- no 64-bit address
                                      // per load in registers), samples past the end of the span or of the segment read as 0 through the range check
        const long   i0 = (seg0 - Kp) * D - (D - 1);
        const long   nrec = n_in - i0 < (long)cnt ? n_in - i0 : (long)cnt;
        const rsrc_t r  = make_rsrc(x + i0, (unsigned)(nrec > 0 ? nrec * 4 : 0));
#pragma unroll
        for (int u = 0; u < NL; ++u) nxt[u] = buf_load_f(r, tid * 4, 256 * u * 4);
    };
    auto store_next = [&]() {
        int mp = mp0, pp = pp0;
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            if (tid + 256 * u < cnt && mp < SEG + Kp) xs[pp * ROW + mp + PAD * (mp >> 4)] = nxt[u];
            mp += q256;
            pp -= r256;
            if (pp < 0) { pp += D; mp += 1; }
        }
    };
    auto stage_direct = [&](long seg0) { // segments too long for the register prefetch (large D)
        int           mp = mp0, pp = pp0;
        constexpr int U = 8; // loads in flight per lane (a load-per-iteration loop pays the memory latency cnt / 256 times)
        for (int s0 = tid; s0 < cnt; s0 += 256 * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) v[u] = sample(seg0, s0 + 256 * u);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (s0 + 256 * u < cnt && mp < SEG + Kp) xs[pp * ROW + mp + PAD * (mp >> 4)] = v[u];
                mp += q256;
                pp -= r256;
                if (pp < 0) { pp += D; mp += 1; }
            }
        }
    };
    const long nseg = (n_out + SEG - 1) / SEG, sfirst = (long)blockIdx.x * kDecimSPW, slast = sfirst + kDecimSPW < nseg ? sfirst + kDecimSPW : nseg;
    if (prefetch && sfirst > 0 && sfirst < slast) load_next(sfirst * SEG);
    const float* pb[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) pb[t] = xs + kq + (16 + PAD) * (16 * (wave * TPW + t) + col);
    const float* pa = tp + 16 + Kp + col - kq;
    for (long sg = sfirst; sg < slast; ++sg) {
        const long seg0 = sg * SEG; // first output of this segment
        if (prefetch && sg > 0) store_next();
        else stage_direct(seg0); // the very first segment of the span reads the carried history in front of x
        __syncthreads();
        if (prefetch && sg + 1 < slast) load_next(seg0 + SEG); // in flight during the MFMA phase below

        f32x4 acc[TPW]; // one 16-block x 16-output tile (256 outputs) each
#pragma unroll
        for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        for (int p = 0; p < D; ++p) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const int   off = 4 * ks + PAD * (ks >> 2);
                const float a   = pa[p * TS - 4 * ks];
#pragma unroll
                for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pb[t][p * ROW + off], acc[t], 0, 0, 0);
            }
        }
#pragma unroll
        for (int t = 0; t < TPW; ++t) {
            const long o = seg0 + 16L * (16 * (wave * TPW + t) + col) + 4 * kq;
            if (o + 3 < n_out) *reinterpret_cast<float4*>(y + o) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
            else
                for (int r = 0; r < 4; ++r)
                    if (o + r < n_out) y[o + r] = acc[t][r];
        }
        __syncthreads(); // every wave is done with the sample rows before the next segment overwrites them
    }
}

// Decimation by 9 and more: the polyphase form above de-interleaves the segment into D phase rows, which stops fitting the LDS at
// D = 11 .. 16, and the register-window kernel behind it spends its time on that de-interleave (D = 16: 104 G input samples/s, 64: 27).  Here the samples stay
// in stream order and the decimation sits in the A operand: with W_i[u] = x[16 i D - Kp + u], u < Kp + 16 D,
//     y[16 i + j] = sum_u A[j][u] W_i[u],   A[j][u] = b[Kp + j D - u]
// -- a band of Kp taps that moves D columns per row, Kp / (Kp + 15 D) of the products non-zero (68 % at K = 32 D), 2 (Kp + 16 D) / D executed flop per input
// sample: HBM-bound.  One tile of 256 outputs (256 D inputs + Kp) per workgroup; the four waves SPLIT the K-steps and their partial tiles are summed
// through LDS.  Bank layout: the 16 columns of a K-step are 16 D floats apart -- two pad floats per 16 D samples put them on different banks (offset of a
// K-step: lane base + wave-uniform part); A's lanes are D floats apart in the tap row -- two pad floats per D taps.
// D = 2^a m with m odd: the pads go in per 2^(a+4) samples and per 2^a taps (shifts, no divisions); the column stride becomes 2 m (2^(a+3) + 1) and the row
// stride 2 m (2^(a-1) + 1) floats -- an odd multiple of 2 either way (a < 2: the tap row needs no padding), so 16 lanes always hit 16 different even banks.
__global__ __launch_bounds__(256) void fir_decim_band_kernel(const float* __restrict__ x, const float* __restrict__ hist /*hist[h] = x[-hcap + h]*/, int hcap,
                                                              const float* __restrict__ tb /*[Kp + 32 D]: index 16 D + t holds b[t]*/, int Kp, float* __restrict__ y, long n_out, long n_in,
                                                              int D, int sha /*ctz(D) + 4*/, int sht /*ctz(D), or 31: no padding*/) {
    const int BLK = 16 * D;
    extern __shared__ float bsm[];
    auto      padx = [sha](int s_) { return s_ + 2 * (s_ >> sha); };
    auto      padt = [sht](int q) { return q + 2 * (q >> sht); };
    const int KS = (Kp + BLK) / 4, NS = 256 * D + Kp, NT = Kp + 32 * D;
    float*    xs   = bsm;                       // padx(NS) + 2
    float*    tl   = xs + padx(NS) + 2;         // padt(NT) + 2
    float*    part = tl + padt(NT) + 2;         // [4 waves][64 lanes][4]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
    const long tile = blockIdx.x, p0 = tile * 256L * D - Kp; // stream position of staged sample 0
    for (int q = tid; q < NT; q += 256) tl[padt(q)] = tb[q];
    constexpr int NB = 24; // loads in flight per lane (D = 16: the whole tile in one round)
    for (int s0 = tid; s0 < NS; s0 += 256 * NB) {
        float v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int  s_ = s0 + 256 * u;
            const long i  = p0 + s_;
            v[u]          = s_ < NS ? (i >= 0 ? (i < n_in ? x[i] : 0.f) : (i >= -(long)hcap ? hist[hcap + i] : 0.f)) : 0.f;
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int s_ = s0 + 256 * u;
            if (s_ < NS) xs[padx(s_)] = v[u];
        }
    }
    __syncthreads();
    using f32x4b = __attribute__((ext_vector_type(4))) float;
    f32x4b      acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = acc0, acc2 = acc0, acc3 = acc0;
    const int   KSw = (KS + 3) / 4, k0 = wave * KSw, k1 = k0 + KSw < KS ? k0 + KSw : KS; // this wave's K-steps
    const float* pb = xs + padx(BLK) * col + kq;            // B: sample 16 D col + u, u = 4 ks + kq  (16 D col has no low bits: the pad count splits exactly)
    const int    qa = Kp + col * D + BLK - kq;              // A: tap-row index of u = kq
    auto step = [&](f32x4b& acc, int ks) {
        const int u = 4 * ks;
        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(tl[padt(qa - u)], pb[u + 2 * (u >> sha)], acc, 0, 0, 0);
    };
    int ks = k0;
    for (; ks + 8 <= k1; ks += 8) { // four accumulators, eight K-steps per round: the operand reads of a round are issued ahead of its MFMAs
        step(acc0, ks); step(acc1, ks + 1); step(acc2, ks + 2); step(acc3, ks + 3);
        step(acc0, ks + 4); step(acc1, ks + 5); step(acc2, ks + 6); step(acc3, ks + 7);
    }
    for (; ks < k1; ++ks) step(acc0, ks);
#pragma unroll
    for (int r = 0; r < 4; ++r) { acc0[r] += acc2[r]; acc1[r] += acc3[r]; }
    *reinterpret_cast<float4*>(part + (wave * 64 + lane) * 4) = make_float4(acc0[0] + acc1[0], acc0[1] + acc1[1], acc0[2] + acc1[2], acc0[3] + acc1[3]);
    __syncthreads();
    // D[row = 4 kq + r][col]: output 16 col + 4 kq + r of the tile; thread o sums the four waves' partial sums of output o
    const int  o = tid, ln = (o >> 4) + 16 * ((o & 15) >> 2), r = o & 3;
    const long m = tile * 256 + o;
    if (m < n_out) y[m] = (part[(0 * 64 + ln) * 4 + r] + part[(1 * 64 + ln) * 4 + r]) + (part[(2 * 64 + ln) * 4 + r] + part[(3 * 64 + ln) * 4 + r]);
}

// tap row of fir_decim_band_kernel: [Kp + 32 D], index 16 D + t holds b[t]; Kp = taps rounded up to a multiple of 4
void fir_decim_band_make_row(const float* taps, size_t ntaps, size_t D, int* Kp_out, std::vector<float>* row) {
    const int Kp = (int)((ntaps + 3) / 4 * 4);
    row->assign((size_t)Kp + 32 * D, 0.f);
    for (size_t t = 0; t < ntaps; ++t) (*row)[16 * D + t] = taps[t];
    *Kp_out = Kp;
}

// y[m] = sum_k b[k] x[m D - k], m < n_out; GR4HIP_UNSUPPORTED when the tile does not fit the LDS
int fir_decim_band_launch(int D, int Kp, const float* x, const float* hist, int hcap, const float* row, float* y, long n_out, long n_in, hipStream_t st) {
    if (D < 2) return GR4HIP_UNSUPPORTED;
    int a = 0;
    while (((D >> a) & 1) == 0) ++a;
    const int    sha = a + 4, sht = a >= 2 ? a : 31;
    const int    NS = 256 * D + Kp, NT = Kp + 32 * D;
    const size_t lds = ((size_t)(NS + 2 * (NS >> sha) + 2) + (size_t)(NT + 2 * (NT >> sht) + 2) + 4 * 64 * 4) * sizeof(float);
    if (lds > 150 * 1024) return GR4HIP_UNSUPPORTED;
    const dim3 grid((unsigned)ceil_div(n_out, 256L));
    auto       kern = fir_decim_band_kernel;
    if (lds > 48 * 1024) GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, hist, hcap, row, Kp, y, n_out, n_in, D, sha, sht);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

// A-fragment table [nch][KS][64] for per-channel taps [nch][ntaps]; Kp = 64 / 128 / 256
void fir_mfma_make_afrag(const float* taps, size_t ntaps, size_t nch, int* Kp_out, int* KS_out, std::vector<float>* af_out) {
    const int Kp = ntaps <= 64 ? 64 : ntaps <= 128 ? 128 : 256, KS = (Kp + 16) / 4;
    af_out->assign(nch * KS * 64, 0.f);
    for (size_t c = 0; c < nch; ++c)
        for (int ks = 0; ks < KS; ++ks)
            for (int l = 0; l < 64; ++l) {
                const int k = Kp + (l & 15) - (4 * ks + (l >> 4)); // tap index b[Kp + j - u]
                (*af_out)[(c * KS + ks) * 64 + l] = (k >= 0 && (size_t)k < ntaps) ? taps[c * ntaps + k] : 0.f;
            }
    *Kp_out = Kp;
    *KS_out = KS;
}

// y[c][i] = sum_k b_c[k] x[c][i - k], i < n; hist[c][Kp] = the Kp samples in front of x[c]; y must be 16-byte aligned, out_stride % 4 == 0
int fir_mfma_launch(int KS, const float* x, long in_stride, const float* hist, const float* afrag, float* y, long out_stride, long n, unsigned nch, hipStream_t st, float* new_hist) {
    const dim3 grid((unsigned)ceil_div(ceil_div(n, (long)kSeg), (long)kSegPerWg), nch);
    switch (KS) {
    case 20: hipLaunchKernelGGL(fir_mfma_kernel<20>, grid, dim3(256), 0, st, x, in_stride, hist, afrag, y, out_stride, n, new_hist); break;
    case 36: hipLaunchKernelGGL(fir_mfma_kernel<36>, grid, dim3(256), 0, st, x, in_stride, hist, afrag, y, out_stride, n, new_hist); break;
    default: hipLaunchKernelGGL(fir_mfma_kernel<68>, grid, dim3(256), 0, st, x, in_stride, hist, afrag, y, out_stride, n, new_hist); break;
    }
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

// y[i] = sum_k b[k] x[i - k] on complex samples; hist = the Kp complex samples in front of x; y must be 16-byte aligned
int fir_mfma_c32_launch(int KS, const float* x, long n, const float* hist, const float* afrag, float* y, hipStream_t st, float* new_hist) {
    const dim3 grid((unsigned)ceil_div(ceil_div(n, (long)kSegC), (long)kSegPerWg));
    const auto xc = reinterpret_cast<const float2*>(x), hc = reinterpret_cast<const float2*>(hist);
    const auto yc = reinterpret_cast<float2*>(y);
    const auto nh = reinterpret_cast<float2*>(new_hist);
    switch (KS) {
    case 20: hipLaunchKernelGGL(fir_mfma_c32_kernel<20>, grid, dim3(256), 0, st, xc, hc, afrag, yc, n, nh); break;
    case 36: hipLaunchKernelGGL(fir_mfma_c32_kernel<36>, grid, dim3(256), 0, st, xc, hc, afrag, yc, n, nh); break;
    default: hipLaunchKernelGGL(fir_mfma_c32_kernel<68>, grid, dim3(256), 0, st, xc, hc, afrag, yc, n, nh); break;
    }
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

// polyphase tap rows [D][Kp + 32]: row p, index 16 + q holds b[q D + p] (zero elsewhere)
void fir_mfma_make_afrag_decim(const float* taps, size_t ntaps, size_t D, int* Kp_out, int* KS_out, std::vector<float>* af_out) {
    const size_t Q  = ceil_div(ntaps, D);
    const int    Kp = Q <= 64 ? 64 : Q <= 128 ? 128 : 256, KS = (Kp + 16) / 4, TS = Kp + 32;
    af_out->assign(D * TS, 0.f);
    for (size_t p = 0; p < D; ++p)
        for (size_t q = 0; q < Q; ++q)
            if (q * D + p < ntaps) (*af_out)[p * TS + 16 + q] = taps[q * D + p];
    *Kp_out = Kp;
    *KS_out = KS;
}

// y[m] = sum_k b[k] x[mD - k], m < n_out; hist = the Kp D samples in front of x; y 16-byte aligned.  Returns GR4HIP_UNSUPPORTED when the
// de-interleaved segment does not fit the LDS.
int fir_mfma_decim_launch(int KS, int D, const float* x, const float* hist, const float* ptaps, float* y, long n_out, hipStream_t st) {
    const int Kp = 4 * KS - 16;
    for (int tpw : {1, 2}) { // smaller segments first: 39 KB of LDS at D = 8 -> four workgroups per CU cover the staging and A-fragment latencies
        const int    seg = 1024 * tpw, row = (seg + Kp) / 16 * 18 + 1;
        const size_t lds = (size_t)D * (row + Kp + 32) * sizeof(float);
        if (lds > 78 * 1024) continue; // two workgroups per CU
        const dim3 grid((unsigned)ceil_div(ceil_div(n_out, (long)seg), (long)kDecimSPW));
#define GR4_DECIM_CASE(KSV, TPWV)                                                                                                         \
    do {                                                                                                                                  \
        auto kern = fir_mfma_decim_kernel<KSV, TPWV>;                                                                                     \
        if (lds > 48 * 1024) GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, hist, ptaps, y, n_out, D);                                                  \
    } while (0)
        if (tpw == 2) { if (KS == 20) GR4_DECIM_CASE(20, 2); else if (KS == 36) GR4_DECIM_CASE(36, 2); else GR4_DECIM_CASE(68, 2); }
        else          { if (KS == 20) GR4_DECIM_CASE(20, 1); else if (KS == 36) GR4_DECIM_CASE(36, 1); else GR4_DECIM_CASE(68, 1); }
#undef GR4_DECIM_CASE
        GR4_LAUNCH_CHECK();
        return GR4HIP_OK;
    }
    return GR4HIP_UNSUPPORTED;
}

} // namespace gr4

using namespace gr4;

namespace gr4 {
void fir_bf16_make_afrag(const float* taps, size_t ntaps, int* KS_out, std::vector<unsigned short>* af, size_t nch, int force_ks); // fir_bf16.hip
int  fir_bf16_launch(int KS, const float* x, long n, const float* hist, int Kh, const void* afrag, float* y, hipStream_t st, float* new_hist, long in_stride, long out_stride, unsigned nch, int delay, int accum);
bool fir_f16_make_afrag(const float* taps, size_t ntaps, int* KS_out, std::vector<unsigned short>* af, size_t nch, int force_ks); // fir_f16.hip
int  fir_f16_launch(int KS, const float* x, long n, const float* hist, int Kh, const void* table, float* y, hipStream_t st, float* new_hist, long in_stride, long out_stride, unsigned nch, int delay, int accum, int guard,
                    unsigned char* flags, long flags_stride, float gthr);
}
struct gr4hip_fir_batched {
    size_t       nch = 0, ntaps = 0;
    int          KS = 0, Kp = 0;
    DeviceBuffer d_afrag, d_hist[2];
    int          cur = 0;
    DeviceBuffer d_bfrag; // > 64 taps: per-channel three-term bf16 tap fragments (fir_bf16.hip)
    int          bfKS = 0;
    DeviceBuffer d_hfrag; // > 32 taps: per-channel two-term f16 tables (fir_f16.hip) -- the default on long spans
    int          hfKS = 0;
    DeviceBuffer d_tapsf, d_flags; // the taps as they are and one byte per channel and segment: what fir_exact_kernel evaluates again behind the f16 launch
    bool         zero_hist = true; // the stream rule (common.hpp): create / reset note it, the next call zeroes d_hist[cur] on its own stream
};

extern "C" {

int gr4hip_fir_batched_create(gr4hip_fir_batched_t** out, size_t nchannels, const float* h_taps, size_t ntaps) {
    GR4_REQUIRE(out && h_taps && nchannels >= 1 && ntaps >= 1, "fir_batched: bad arguments");
    if (ntaps > 256 || nchannels > 65535) { set_error("fir_batched: device path supports ntaps <= 256 and <= 65535 channels (got %zu taps, %zu channels)", ntaps, nchannels); return GR4HIP_UNSUPPORTED; }
    auto* f = new (std::nothrow) gr4hip_fir_batched();
    GR4_REQUIRE(f, "out of host memory");
    f->nch   = nchannels;
    f->ntaps = ntaps;
    std::vector<float> af;
    fir_mfma_make_afrag(h_taps, ntaps, nchannels, &f->Kp, &f->KS, &af);
    int rc = f->d_afrag.ensure(af.size() * sizeof(float));
    if (!rc) { hipError_t e = upload_fresh(f->d_afrag.ptr, af.data(), af.size() * sizeof(float)); if (e != hipSuccess) { set_error("fir_batched: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
    static const size_t kBfMinTaps = [] { const char* e = std::getenv("GR4HIP_FIR_BATCHED_BF16_MIN_TAPS"); return e ? (size_t)std::atoi(e) : (size_t)33; }(); // developer knob
    if (!rc && ntaps >= kBfMinTaps && ntaps > 32) {
        std::vector<unsigned short> bf;
        fir_bf16_make_afrag(h_taps, ntaps, &f->bfKS, &bf, nchannels, 0);
        rc = f->d_bfrag.ensure(bf.size() * sizeof(unsigned short));
        if (!rc) { hipError_t e = upload_fresh(f->d_bfrag.ptr, bf.data(), bf.size() * sizeof(unsigned short)); if (e != hipSuccess) { set_error("fir_batched: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
    }
    if (!rc && ntaps >= kBfMinTaps && ntaps > 32) {
        std::vector<unsigned short> hf;
        if (fir_f16_make_afrag(h_taps, ntaps, &f->hfKS, &hf, nchannels, 0)) {
            rc = f->d_hfrag.ensure(hf.size() * sizeof(unsigned short));
            if (!rc) { hipError_t e = upload_fresh(f->d_hfrag.ptr, hf.data(), hf.size() * sizeof(unsigned short)); if (e != hipSuccess) { set_error("fir_batched: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
        } else f->hfKS = 0;
        if (!rc && f->hfKS) {
            rc = f->d_tapsf.ensure(nchannels * ntaps * sizeof(float));
            if (!rc) { hipError_t e = upload_fresh(f->d_tapsf.ptr, h_taps, nchannels * ntaps * sizeof(float)); if (e != hipSuccess) { set_error("fir_batched: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
        }
    }
    for (int k = 0; k < 2 && !rc; ++k) rc = f->d_hist[k].ensure(nchannels * f->Kp * sizeof(float));
    if (!rc) rc = gr4hip_fir_batched_reset(f);
    if (rc) { delete f; return rc; }
    *out = f;
    return GR4HIP_OK;
}

int gr4hip_fir_batched_reset(gr4hip_fir_batched_t* f) {
    GR4_REQUIRE(f, "fir_batched_reset: null handle");
    f->zero_hist = true; // (a launch still in flight on the caller's stream may be writing the other half of the pair: the zeroing goes behind it, on the next call's stream)
    return GR4HIP_OK;
}

int gr4hip_fir_batched_process(gr4hip_fir_batched_t* f, const float* d_in, size_t in_stride, size_t n, float* d_out, size_t out_stride, gr4hip_stream_t stream) {
    GR4_REQUIRE(f, "fir_batched_process: null handle");
    if (n == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in && d_out && in_stride >= n && out_stride >= n, "fir_batched_process: null pointer or stride shorter than n");
    GR4_REQUIRE(((uintptr_t)d_out % 16 == 0) && (out_stride % 4 == 0), "fir_batched_process: output must be 16-byte aligned with a stride multiple of 4");
    hipStream_t st = as_stream(stream);
    if (f->zero_hist) {
        GR4_HIP_TRY(hipMemsetAsync(f->d_hist[f->cur].ptr, 0, f->nch * f->Kp * sizeof(float), st));
        f->zero_hist = false;
    }
    const float *hist = (const float*)f->d_hist[f->cur].ptr, *af = (const float*)f->d_afrag.ptr;
    int rc;
    if (f->hfKS && n >= 32768 && (uintptr_t)d_in % 16 == 0 && in_stride % 4 == 0 && !dev_switch(kDevFirNoBf16x3) && !dev_switch(kDevFirNoF16x2)) { // two-term f16 form (fir_f16.hip): same history layout
        const long nsegs = (long)ceil_div(n, (size_t)4096);
        rc = f->d_flags.ensure((size_t)nsegs * f->nch);
        if (!rc) rc = fir_f16_launch(f->hfKS, d_in, (long)n, hist, f->Kp, f->d_hfrag.ptr, d_out, st, nullptr, (long)in_stride, (long)out_stride, (unsigned)f->nch, 0, 0, 1, (unsigned char*)f->d_flags.ptr, nsegs, 0.f);
        if (!rc) rc = fir_exact_launch(d_in, (long)n, hist, f->Kp, (const float*)f->d_tapsf.ptr, (int)f->ntaps, 1, 0, d_out, (long)n, (const unsigned char*)f->d_flags.ptr, 12, nullptr, st, (unsigned)f->nch,
                                       (long)in_stride, (long)out_stride, (long)f->ntaps, nsegs); // the marked segments again on the FP64 matrix pipe
    } else if (f->bfKS && n >= 32768 && (uintptr_t)d_in % 16 == 0 && in_stride % 4 == 0 && !dev_switch(kDevFirNoBf16x3)) // three-term bf16 form (fir_bf16.hip): same history layout
        rc = fir_bf16_launch(f->bfKS, d_in, (long)n, hist, f->Kp, f->d_bfrag.ptr, d_out, st, nullptr, (long)in_stride, (long)out_stride, (unsigned)f->nch, 0, 0);
    else
        rc = fir_mfma_launch(f->KS, d_in, (long)in_stride, hist, af, d_out, (long)out_stride, (long)n, (unsigned)f->nch, st, nullptr);
    if (rc) return rc;
    hipLaunchKernelGGL(fir_batched_hist_kernel, dim3((unsigned)ceil_div(f->Kp, 64), (unsigned)f->nch), dim3(64), 0, st, d_in, (long)in_stride, hist,
                       (float*)f->d_hist[f->cur ^ 1].ptr, (long)n, f->Kp);
    GR4_LAUNCH_CHECK();
    f->cur ^= 1;
    return GR4HIP_OK;
}

int gr4hip_fir_batched_destroy(gr4hip_fir_batched_t* f) { delete f; return GR4HIP_OK; }

} // extern "C"
