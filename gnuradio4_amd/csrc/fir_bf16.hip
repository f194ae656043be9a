// fir_bf16.hip -- fir_filter<float>, 65 .. 256 taps, on the bf16 matrix pipe with float32 accuracy (three-term splits).
//
// The block-Toeplitz contraction of fir_batched.hip is bound by v_mfma_f32_16x16x4_f32, which runs at the FP32 VALU's rate (and, measured in chain_td.hip, on
// its issue slot): 256 taps = 251 Gsamples/s = 2 TB/s, a quarter of the HBM roofline.  The bf16 MFMA is 15x faster per flop.  A float32 value is the exact sum
// of three bf16 values (8 significant bits each: x = h + m + l with h = bf16(x), m = bf16(x - h), l = bf16(x - h - m); the residual is <= 2^-25 |x|), bf16 x
// bf16 products are exact in float32, and the MFMA accumulates in float32.  With samples AND taps split, x b = sum_{i,j} x_i b_j; the six terms with
// i + j <= 2 (hh, hm, mh, hl, lh, mm) carry everything above 2^-23 |x b| -- the size of float32's own rounding of that product.  Six bf16 MFMAs replace the
// 8 f32 MFMAs of the same K range (16x16x32 against 16x16x4): 102 against 256 matrix-pipe cycles.
//
//     y[16 i + j] = sum_u A[j][u] B[u][i],   B[u][i] = x[16 i - Hb + u],   A[j][u] = b[Hb + j - u],   u < Kw = 32 KS,   Hb = Kw - 16 >= taps - 1
//
// The samples are split once, while they are staged (three bf16 planes in LDS; a lane's B operand is 8 consecutive samples = one ds_read_b128 per plane),
// the taps on the host (three fragment tables, resident in registers).  The planes are NOT padded: ds_read_b128 serves a wave in four fixed groups of 16 lanes, and
// with lane (col, kq) reading chunk 2 col + kq every group covers 16 different chunks mod 16 -- the first version padded 8 elements per 128 "against" conflicts and
// had 62 % of its LDS cycles in conflicts (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, LDS 77 % busy).  4096 outputs per segment, two tiles (four accumulators) in flight per wave, the
// next segment requested into registers before the MFMAs, workgroup 0 writes the next history.  Bound: HBM (8 B per sample) once the matrix pipe is out of
// the way.  Parity: the same 1e-5 bar as every float32 path; measured error against the float64 oracle ~2e-7, like the f32 MFMA kernel's.
// (Measured and dropped: the m and l tap planes' fragments in LDS instead of registers -- 180 instead of 228 registers at 256 taps, one more LDS read per K-step and
// tile pair: 303 instead of 326 Gsamples/s.  With all three planes resident the compiler re-uses one B register quad and waits for the LDS three times per K-step;
// the extra reads cost more than those waits.)
//
// Round 3, windows of 256 / 288 samples (KS = 8, 9: 210 .. 256 taps, and the 256-tap slices of longer filters): fir_mfma_bf16x3_shared_kernel.  At KS = 9 the kernel
// above reads 0.5 KB of B operand from LDS per MFMA -- as many LDS cycles as matrix-pipe cycles -- and holds the package at its 1400 W cap with the shader clock at
// 1.77 GHz (tools/power_probe.sh).  The second kernel maps outputs to tiles so that four tiles of a wave walk ONE fragment stream (a third of the operand
// reads), double-buffers the planes so that the next segment is split beside the MFMAs, and keeps two segments of loads in flight: 256 taps 360 -> 389 Gsamples/s,
// BASELINE configs[3] (64 channels x 256 taps) 357 -> 387, 512 taps 134 -> 173, 1024 taps 56 -> 81 (same box, tools/ab_run.sh tools/fir_ab.py).  Narrower windows
// are HBM- and power-bound, not operand-bound, and run 0 .. 15 % FASTER on the first kernel (its stores are whole 256-byte rows; the tile map of the second one
// leaves 64-byte pieces: +7 % when timed with row stores, -DGR4_T_COALESCED_STORE), so they stay there (kBfSharedMinKS).
// (The complex kernel below was rebuilt the same way -- 4096 complex outputs per segment, six planes double-buffered, sixteen accumulators per wave -- and measured
// 163-164 against 168 Gsamples/s at 256 taps, 175 against 179 at 224: it is at the same power cap with two streams' worth of MFMAs per sample, and stays as it is.)
#include "common.hpp"
#include "buffer_ops.hpp"
#include "ewise.hpp"
#include "fir_f16_common.hpp" // (hf_wave_sum)
#include "fir_band_hooks.hpp"

#include <algorithm>
#include <cstring>

namespace gr4 {

using bf16x8  = __attribute__((ext_vector_type(8))) __bf16;
using f32x4_b = __attribute__((ext_vector_type(4))) float;
using u32x4_b = __attribute__((ext_vector_type(4))) unsigned;

#ifndef GR4_BF16_WG_PER_CU
#define GR4_BF16_WG_PER_CU 2
#endif
#ifndef GR4_BF16_TARGET_WGS
#define GR4_BF16_TARGET_WGS 512 // two workgroups per CU, each walking up to 128 segments (measured against 1024 x 64, 2048 x 32, 4096 x 16: 387 / 386 / 380 / 371 Gsamples/s at 256 taps)
#define GR4_BF16_MAX_SPW 128
#endif
constexpr int kBfSeg = 4096, kBfSegPerWg = 4;
constexpr int kBfSharedMinKS = 8; // measured: 8 % faster at 256 taps, equal at 200, slower below (those windows are HBM- and power-bound, not operand-bound)

using bf16x2_b = __attribute__((ext_vector_type(2))) __bf16;
using f32x2_b  = __attribute__((ext_vector_type(2))) float;
// two samples -> their three bf16 terms h, m, l, each pair packed in one dword (v_cvt_pk_bf16_f32: round to nearest even; the residuals are exact in float32)
__device__ __forceinline__ void bf_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const f32x2_b  v  = {x0, x1};
    const bf16x2_b hh = __builtin_convertvector(v, bf16x2_b);
    const f32x2_b  r1 = v - __builtin_convertvector(hh, f32x2_b);
    const bf16x2_b mm = __builtin_convertvector(r1, bf16x2_b);
    const f32x2_b  r2 = r1 - __builtin_convertvector(mm, f32x2_b);
    const bf16x2_b ll = __builtin_convertvector(r2, bf16x2_b);
    h = __builtin_bit_cast(unsigned, hh);
    m = __builtin_bit_cast(unsigned, mm);
    l = __builtin_bit_cast(unsigned, ll);
}

template <int KS> // K-steps of 32: window Kw = 32 KS, Hb = Kw - 16 samples in front of a 16-output block
__global__ __launch_bounds__(256) void fir_mfma_bf16x3_kernel(const float* __restrict__ x0, const float* __restrict__ hist0 /*the Kh samples in front of x*/, int Kh,
                                                               const u32x4_b* __restrict__ afrag0 /*[3 planes][KS][64 lanes]: 8 bf16 each*/, float* __restrict__ y0, long n,
                                                               float* __restrict__ new_hist, long in_stride, long out_stride /*channel blockIdx.y: x0 + c in_stride, hist0 + c Kh, afrag0 + c 3 KS 64, y0 + c out_stride*/,
                                                               int delay /*this pass filters x delayed by `delay` samples (a multiple of 16) ...*/, int accum /*... and adds to y: filters longer than 256 taps run as slices*/) {
    const float*   x     = x0 + (long)blockIdx.y * in_stride;
    const float*   hist  = hist0 + (long)blockIdx.y * Kh;
    const u32x4_b* afrag = afrag0 + (long)blockIdx.y * 3 * KS * 64;
    float*         y     = y0 + (long)blockIdx.y * out_stride;
    constexpr int Kw = 32 * KS, Hb = Kw - 16, NS = kBfSeg + Hb; // staged samples per segment (a multiple of 16)
    constexpr int PL  = NS + 8;                                   // bf16 elements per plane
    constexpr int NL4 = (NS / 4 + 255) / 256;                   // float4 loads a lane holds for the next segment
    __shared__ __attribute__((aligned(16))) unsigned short pl[3 * PL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
    auto      P   = [](int s_) { return s_; }; // no padding: lane (col, kq) reads the 16-byte chunk 2 col + kq (+ 4 ks) -- ds_read_b128's four lane groups each cover 16 different chunks mod 16

    u32x4_b a[3][KS]; // A fragments of the three tap planes
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[p][ks] = afrag[(p * KS + ks) * 64 + lane];

    float4 nxt[NL4];
    auto   load_next = [&](long seg0) { // seg0 >= kBfSeg > Hb: nothing below 0; past the end of the span / of the segment the range check returns 0
        const long   i0   = seg0 - Hb - delay; // (seg0 >= kBfSeg > Hb + delay is NOT guaranteed for delayed passes: they stage every segment through the general path below)
        const long   nrec = n - i0 < (long)NS ? n - i0 : (long)NS;
        const rsrc_t r    = make_rsrc(x + i0, (unsigned)(nrec > 0 ? nrec * 4 : 0));
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, 256 * u * 16, 0);
            nxt[u]       = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto put4 = [&](int q, float4 v) { // samples 4 q .. 4 q + 3 of the staged range -> the three planes
        unsigned h0, m0, l0, h1, m1, l1;
        bf_split2(v.x, v.y, h0, m0, l0);
        bf_split2(v.z, v.w, h1, m1, l1);
        const int e = P(4 * q); // 4 consecutive elements never straddle a pad (pads sit at multiples of 128)
        *reinterpret_cast<uint2*>(pl + e)          = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(pl + PL + e)     = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(pl + 2 * PL + e) = make_uint2(l0, l1);
    };
    const long nseg = (n + kBfSeg - 1) / kBfSeg, sfirst = (long)blockIdx.x * kBfSegPerWg, slast = sfirst + kBfSegPerWg < nseg ? sfirst + kBfSegPerWg : nseg;
    const bool pre = delay == 0; // register prefetch of the next segment
    if (pre && sfirst > 0 && sfirst < slast) load_next(sfirst * kBfSeg);
    for (long sg = sfirst; sg < slast; ++sg) {
        const long seg0 = sg * kBfSeg;
        if (pre && sg > 0) {
#pragma unroll
            for (int u = 0; u < NL4; ++u) {
                const int q = tid + 256 * u;
                if (q < NS / 4) put4(q, nxt[u]);
            }
        } else {
            for (int q = tid; q < NS / 4; q += 256) { // the first segment of the span reads the carried history in front of x
                float t[4];
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const long i = seg0 + 4L * q + c - Hb - delay;
                    t[c]         = i >= 0 ? (i < n ? x[i] : 0.f) : (i >= -(long)Kh ? hist[Kh + i] : 0.f);
                }
                put4(q, make_float4(t[0], t[1], t[2], t[3]));
            }
        }
        __syncthreads();
        if (pre && sg + 1 < slast) load_next(seg0 + kBfSeg); // in flight during the MFMAs below
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) {
            const int ib0 = 16 * (4 * wave + 2 * pair), ib1 = ib0 + 16; // first 16-sample block of each tile
            f32x4_b   c0 = {0.f, 0.f, 0.f, 0.f}, d0 = c0, c1 = c0, d1 = c0; // per tile: hh + hm + mh, and hl + lh + mm
            const int s0 = 16 * (ib0 + col) + 8 * kq, s1 = 16 * (ib1 + col) + 8 * kq;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const unsigned short* q0 = pl + P(s0 + 32 * ks);
                const unsigned short* q1 = pl + P(s1 + 32 * ks);
                const bf16x8 bh0 = *reinterpret_cast<const bf16x8*>(q0), bm0 = *reinterpret_cast<const bf16x8*>(q0 + PL), bl0 = *reinterpret_cast<const bf16x8*>(q0 + 2 * PL);
                const bf16x8 bh1 = *reinterpret_cast<const bf16x8*>(q1), bm1 = *reinterpret_cast<const bf16x8*>(q1 + PL), bl1 = *reinterpret_cast<const bf16x8*>(q1 + 2 * PL);
                const bf16x8 ah = __builtin_bit_cast(bf16x8, a[0][ks]), am = __builtin_bit_cast(bf16x8, a[1][ks]), al = __builtin_bit_cast(bf16x8, a[2][ks]);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh1, c1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl0, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl1, d1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm1, c1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh0, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh1, d1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh0, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh1, c1, 0, 0, 0);
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm0, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm1, d1, 0, 0, 0);
            }
            // D[row = 4 kq + r][col]: y[16 (ib + col) + 4 kq + r]; the small terms are added to the large ones last
            const long o0 = seg0 + 16L * (ib0 + col) + 4 * kq, o1 = seg0 + 16L * (ib1 + col) + 4 * kq;
            auto outp = [&](long o, const f32x4_b& c, const f32x4_b& d) {
                if (o + 3 < n) {
                    float4 v = make_float4(c[0] + d[0], c[1] + d[1], c[2] + d[2], c[3] + d[3]);
                    if (accum) { const float4 p = *reinterpret_cast<const float4*>(y + o); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
                    *reinterpret_cast<float4*>(y + o) = v;
                } else {
                    for (int r = 0; r < 4; ++r)
                        if (o + r < n) y[o + r] = (accum ? y[o + r] : 0.f) + (c[r] + d[r]);
                }
            };
            outp(o0, c0, d0);
            outp(o1, c1, d1);
        }
        __syncthreads(); // every wave is done with the staged segment before the next one overwrites it
    }
    if (new_hist != nullptr && blockIdx.x == 0 && blockIdx.y == 0) { // the other half of the caller's ping-pong pair: nobody reads it in this launch
        for (int h = tid; h < Kh; h += 256) {
            const long i = n - Kh + h;
            new_hist[h]  = i >= 0 ? x[i] : hist[Kh + i];
        }
    }
}

// Output (column i, tile t, row j) of a segment = y[seg0 + 256 i + 16 t + j]: the 16 columns of a tile are 256 samples apart, and tile t + 2 sits 32 samples = one
// K-step behind tile t, so B_{t+2}(ks - 1) IS B_t(ks), lane for lane.  A wave owns four tiles tb, tb + 2, tb + 4, tb + 6 and walks ONE fragment stream
// F(m) = staged[256 col + 16 tb + 32 m + 8 kq ..+8), m < KS + 3; tile j uses F(m) as its K-step m - j: KS + 3 fragment reads feed 4 KS K-steps (a third of the
// LDS operand traffic of a read per tile and K-step at KS = 9).  The planes are double-buffered: the next segment is split and written BESIDE the MFMAs
// (its VALU and ds_write instructions fill the matrix pipe's issue gaps), one barrier per segment.
template <int KS> // K-steps of 32: window Kw = 32 KS, Hb = Kw - 16 samples in front of a 16-output block
__global__ __launch_bounds__(256, GR4_BF16_WG_PER_CU) void fir_mfma_bf16x3_shared_kernel(const float* __restrict__ x0, const float* __restrict__ hist0 /*the Kh samples in front of x*/, int Kh,
                                                               const u32x4_b* __restrict__ afrag0 /*[3 planes][KS][64 lanes]: 8 bf16 each*/, float* __restrict__ y0, long n,
                                                               float* __restrict__ new_hist, long in_stride, long out_stride /*channel blockIdx.y: x0 + c in_stride, hist0 + c Kh, afrag0 + c 3 KS 64, y0 + c out_stride*/,
                                                               int delay /*this pass filters x delayed by `delay` samples (a multiple of 16) ...*/, int accum /*... and adds to y: filters longer than 256 taps run as slices*/,
                                                               int seg_per_wg) {
    const float*   x     = x0 + (long)blockIdx.y * in_stride;
    const float*   hist  = hist0 + (long)blockIdx.y * Kh;
    const u32x4_b* afrag = afrag0 + (long)blockIdx.y * 3 * KS * 64;
    float*         y     = y0 + (long)blockIdx.y * out_stride;
    constexpr int Kw = 32 * KS, Hb = Kw - 16, NS = kBfSeg + Hb; // staged samples per segment (a multiple of 16)
    constexpr int PL  = NS + 8 * (NS / 256) + 16;               // bf16 elements per plane: one 16-byte chunk of padding per 256 samples (see P below)
    constexpr int NL4 = (NS / 4 + 255) / 256;                   // float4 loads a lane holds for the next segment
    constexpr int NM  = KS + 3;                                 // fragments of a wave's stream
    __shared__ __attribute__((aligned(16))) unsigned short pls[2][3 * PL];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
    // lane (col, kq) reads the 16-byte chunk 32 col + kq (+ 4 m): columns are 256 samples apart, so one chunk of padding per 256 samples puts the 16 columns of a
    // ds_read_b128 lane group on 16 different chunks mod 16
    auto P = [](int s_) { return s_ + 8 * (s_ >> 8); };

    u32x4_b a[3][KS]; // A fragments of the three tap planes
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[p][ks] = afrag[(p * KS + ks) * 64 + lane];

    float4 nxa[NL4], nxb[NL4]; // two segments ahead: one set is being split into the other plane buffer while the loads of the other are in flight
    auto   load_next = [&](float4 (&nxt)[NL4], long seg0) { // seg0 >= kBfSeg > Hb: nothing below 0; past the end of the span / of the segment the range check returns 0
        const long   i0   = seg0 - Hb - delay; // >= 0 from the second segment on (Hb + delay <= kBfSeg: the launcher checks)
        const long   nrec = n - i0 < (long)NS ? n - i0 : (long)NS;
        const rsrc_t r    = make_rsrc(x + i0, (unsigned)(nrec > 0 ? nrec * 4 : 0));
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, 256 * u * 16, 0);
            nxt[u]       = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto put4e = [&](unsigned short* pl, int e, float4 v) { // four samples -> elements e .. e + 3 of the three planes
        unsigned h0, m0, l0, h1, m1, l1;
        bf_split2(v.x, v.y, h0, m0, l0);
        bf_split2(v.z, v.w, h1, m1, l1);
        *reinterpret_cast<uint2*>(pl + e)          = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(pl + PL + e)     = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(pl + 2 * PL + e) = make_uint2(l0, l1);
    };
    auto put4 = [&](unsigned short* pl, int q, float4 v) { put4e(pl, P(4 * q), v); }; // samples 4 q .. 4 q + 3 of the staged range (4 consecutive elements never straddle a pad: pads sit at multiples of 256)
    auto put_next = [&](unsigned short* pl, int u, const float4& v) { // branch-free (it sits between MFMAs): lanes past the staged range write into the spare elements behind each plane
        const int q = tid + 256 * u;
        put4e(pl, (256 * (u + 1) <= NS / 4 || q < NS / 4) ? P(4 * q) : PL - 16 + 4 * (lane & 3), v);
    };
    auto stage_general = [&](unsigned short* pl, long seg0) { // any segment, sample by sample: the carried history in front of x, delayed passes, zeros past the end
        for (int q = tid; q < NS / 4; q += 256) {
            float t[4];
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const long i = seg0 + 4L * q + c - Hb - delay;
                t[c]         = i >= 0 ? (i < n ? x[i] : 0.f) : (i >= -(long)Kh ? hist[Kh + i] : 0.f);
            }
            put4(pl, q, make_float4(t[0], t[1], t[2], t[3]));
        }
    };
    const long nseg = (n + kBfSeg - 1) / kBfSeg, sfirst = (long)blockIdx.x * seg_per_wg, slast = sfirst + seg_per_wg < nseg ? sfirst + seg_per_wg : nseg;
    if (sfirst < slast) {
        if (sfirst > 0) {
            load_next(nxa, sfirst * kBfSeg);
#pragma unroll
            for (int u = 0; u < NL4; ++u) put_next(pls[0], u, nxa[u]);
        } else {
            stage_general(pls[0], sfirst * kBfSeg);
        }
        load_next(nxa, (sfirst + 1) * kBfSeg); // (past the end of the span: empty loads)
    }
    __syncthreads();
    const int tb = (wave >> 1) + 8 * (wave & 1); // waves 0, 1: the even tiles from 0 / 8; waves 2, 3: the odd tiles from 1 / 9
    const int sb = 256 * col + 16 * tb + 8 * kq;
    // one segment: MFMAs on `pl`; `cur` (the segment behind it, requested a segment ago) goes into `plo` meanwhile; the segment behind THAT is requested into `oth` first
    auto segment = [&](long sg, const unsigned short* pl, unsigned short* plo, float4 (&cur)[NL4], float4 (&oth)[NL4]) {
        const long seg0 = sg * kBfSeg;
        load_next(oth, seg0 + 2 * kBfSeg);
        f32x4_b c[4], d[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) c[j] = d[j] = f32x4_b{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int m = 0; m < NM; ++m) {
            const unsigned short* q = pl + P(sb + 32 * m);
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(q), bm = *reinterpret_cast<const bf16x8*>(q + PL), bl = *reinterpret_cast<const bf16x8*>(q + 2 * PL);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int ks = m - j;
                if (ks < 0 || ks >= KS) continue;
                const bf16x8 ah = __builtin_bit_cast(bf16x8, a[0][ks]), am = __builtin_bit_cast(bf16x8, a[1][ks]), al = __builtin_bit_cast(bf16x8, a[2][ks]);
                c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c[j], 0, 0, 0);
                d[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, d[j], 0, 0, 0);
                c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c[j], 0, 0, 0);
                d[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, d[j], 0, 0, 0);
                c[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c[j], 0, 0, 0);
                d[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, d[j], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < NL4; ++u) // the next segment's samples are split and written into the other buffer beside the MFMAs, spread over the stream
                if (u * NM / NL4 == m) put_next(plo, u, cur[u]);
        }
        // D[row = 4 kq + r][col]: y[seg0 + 256 col + 16 t + 4 kq + r]; the small terms are added to the large ones last
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#ifdef GR4_T_COALESCED_STORE // timing only (wrong positions): what the 64-byte store pieces cost
            const long o = seg0 + (wave * 4 + j) * 256 + lane * 4;
#else
            const long o = seg0 + 256L * col + 16 * (tb + 2 * j) + 4 * kq;
#endif
            if (o + 3 < n) {
                float4 v = make_float4(c[j][0] + d[j][0], c[j][1] + d[j][1], c[j][2] + d[j][2], c[j][3] + d[j][3]);
                if (accum) { const float4 p = *reinterpret_cast<const float4*>(y + o); v.x += p.x; v.y += p.y; v.z += p.z; v.w += p.w; }
                *reinterpret_cast<float4*>(y + o) = v;
            } else {
                for (int r = 0; r < 4; ++r)
                    if (o + r < n) y[o + r] = (accum ? y[o + r] : 0.f) + (c[j][r] + d[j][r]);
            }
        }
        __syncthreads(); // the other buffer is complete, and every wave is done with this one before the segment after the next overwrites it
    };
    for (long sg = sfirst; sg < slast; sg += 2) {
        segment(sg, pls[0], pls[1], nxa, nxb);
        if (sg + 1 < slast) segment(sg + 1, pls[1], pls[0], nxb, nxa);
    }
    if (new_hist != nullptr && blockIdx.x == 0 && blockIdx.y == 0) { // the other half of the caller's ping-pong pair: nobody reads it in this launch
        for (int h = tid; h < Kh; h += 256) {
            const long i = n - Kh + h;
            new_hist[h]  = i >= 0 ? x[i] : hist[Kh + i];
        }
    }
}

// fir_filter<complex<float>> (real taps on interleaved {re, im} samples) on the same scheme: the two components are two real streams under the same taps -- six
// planes (re / im x h / m / l), 2048 complex outputs per segment, a wave's two tiles with re and im accumulators each (eight accumulators in flight), D leaves
// re-interleaved as two 16-byte stores per lane.  hist: the Kh complex samples in front of x.
constexpr int kBfSegC = 2048;
template <int KS>
__global__ __launch_bounds__(256) void fir_mfma_bf16x3_c32_kernel(const float2* __restrict__ x, const float2* __restrict__ hist, int Kh, const u32x4_b* __restrict__ afrag,
                                                                   float2* __restrict__ y, long n, float2* __restrict__ new_hist) {
    constexpr int Kw = 32 * KS, Hb = Kw - 16, NS = kBfSegC + Hb;
    constexpr int PL  = NS + 8;
    constexpr int NL4 = (NS / 2 + 255) / 256; // float4 loads (two complex samples each) a lane holds for the next segment
    __shared__ __attribute__((aligned(16))) unsigned short pl[6 * PL]; // re h, m, l, then im h, m, l
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
    auto      P   = [](int s_) { return s_; }; // no padding: lane (col, kq) reads the 16-byte chunk 2 col + kq (+ 4 ks) -- ds_read_b128's four lane groups each cover 16 different chunks mod 16
    u32x4_b   a[3][KS];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[p][ks] = afrag[(p * KS + ks) * 64 + lane];
    float4 nxt[NL4];
    auto   load_next = [&](long seg0) {
        const long   i0   = seg0 - Hb;
        const long   nrec = n - i0 < (long)NS ? n - i0 : (long)NS;
        const rsrc_t r    = make_rsrc(x + i0, (unsigned)(nrec > 0 ? nrec * 8 : 0));
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, 256 * u * 16, 0);
            nxt[u]       = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto put2 = [&](int q, float4 v) { // complex samples 2 q, 2 q + 1 of the staged range -> the six planes
        unsigned h, m, l;
        const int e = P(2 * q);
        bf_split2(v.x, v.z, h, m, l);
        *reinterpret_cast<unsigned*>(pl + e)          = h;
        *reinterpret_cast<unsigned*>(pl + PL + e)     = m;
        *reinterpret_cast<unsigned*>(pl + 2 * PL + e) = l;
        bf_split2(v.y, v.w, h, m, l);
        *reinterpret_cast<unsigned*>(pl + 3 * PL + e) = h;
        *reinterpret_cast<unsigned*>(pl + 4 * PL + e) = m;
        *reinterpret_cast<unsigned*>(pl + 5 * PL + e) = l;
    };
    const long nseg = (n + kBfSegC - 1) / kBfSegC, sfirst = (long)blockIdx.x * kBfSegPerWg, slast = sfirst + kBfSegPerWg < nseg ? sfirst + kBfSegPerWg : nseg;
    if (sfirst > 0 && sfirst < slast) load_next(sfirst * kBfSegC);
    for (long sg = sfirst; sg < slast; ++sg) {
        const long seg0 = sg * kBfSegC;
        if (sg > 0) {
#pragma unroll
            for (int u = 0; u < NL4; ++u) {
                const int q = tid + 256 * u;
                if (q < NS / 2) put2(q, nxt[u]);
            }
        } else {
            for (int q = tid; q < NS / 2; q += 256) { // the first segment of the span reads the carried history in front of x
                float2 t[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const long i = 2L * q + c - Hb;
                    t[c]         = i >= 0 ? (i < n ? x[i] : make_float2(0.f, 0.f)) : (i >= -(long)Kh ? hist[Kh + i] : make_float2(0.f, 0.f));
                }
                put2(q, make_float4(t[0].x, t[0].y, t[1].x, t[1].y));
            }
        }
        __syncthreads();
        if (sg + 1 < slast) load_next(seg0 + kBfSegC);
        {
            const int ib0 = 16 * (2 * wave), ib1 = ib0 + 16; // this wave's two tiles
            f32x4_b   cr0 = {0.f, 0.f, 0.f, 0.f}, dr0 = cr0, ci0 = cr0, di0 = cr0, cr1 = cr0, dr1 = cr0, ci1 = cr0, di1 = cr0;
            const int s0 = 16 * (ib0 + col) + 8 * kq, s1 = 16 * (ib1 + col) + 8 * kq;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const unsigned short* q0 = pl + P(s0 + 32 * ks);
                const unsigned short* q1 = pl + P(s1 + 32 * ks);
                const bf16x8 ah = __builtin_bit_cast(bf16x8, a[0][ks]), am = __builtin_bit_cast(bf16x8, a[1][ks]), al = __builtin_bit_cast(bf16x8, a[2][ks]);
                auto six = [&](const unsigned short* qp, f32x4_b& c, f32x4_b& d) { // hh, hm, mh -> c; hl, lh, mm -> d
                    const bf16x8 bh = *reinterpret_cast<const bf16x8*>(qp), bm = *reinterpret_cast<const bf16x8*>(qp + PL), bl = *reinterpret_cast<const bf16x8*>(qp + 2 * PL);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, d, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, d, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, d, 0, 0, 0);
                };
                six(q0, cr0, dr0);
                six(q0 + 3 * PL, ci0, di0);
                six(q1, cr1, dr1);
                six(q1 + 3 * PL, ci1, di1);
            }
            auto out2 = [&](int ib, const f32x4_b& cr, const f32x4_b& dr, const f32x4_b& ci, const f32x4_b& di) { // D[row = 4 kq + r][col]: y[16 (ib + col) + 4 kq + r]
                const long o = seg0 + 16L * (ib + col) + 4 * kq;
                if (o + 3 < n) {
                    float4* d = reinterpret_cast<float4*>(y + o);
                    d[0]      = make_float4(cr[0] + dr[0], ci[0] + di[0], cr[1] + dr[1], ci[1] + di[1]);
                    d[1]      = make_float4(cr[2] + dr[2], ci[2] + di[2], cr[3] + dr[3], ci[3] + di[3]);
                } else {
                    for (int r = 0; r < 4; ++r)
                        if (o + r < n) y[o + r] = make_float2(cr[r] + dr[r], ci[r] + di[r]);
                }
            };
            out2(ib0, cr0, dr0, ci0, di0);
            out2(ib1, cr1, dr1, ci1, di1);
        }
        __syncthreads();
    }
    if (new_hist != nullptr && blockIdx.x == 0) {
        for (int h = tid; h < Kh; h += 256) {
            const long i = n - Kh + h;
            new_hist[h]  = i >= 0 ? x[i] : hist[Kh + i];
        }
    }
}


// Decimating fir_filter<float> with short polyphase branches (decimation 2 .. 9, window Hb + 15 D + 1 <= 288 samples): the band form of fir_decim_band_kernel
// (samples in stream order, A[j][u] = b[Hb + j D - u]: the decimation sits in the A operand) with the three-term products -- the polyphase kernel on the f32 MFMA
// is bound by that instruction (D = 2, 256 taps: 306 G input samples/s).  1024 outputs (1024 D inputs) per segment, one tile per wave, one accumulator per term.
constexpr int kBdSegOut = 1024, kBdMaxNL4 = 10;
template <int KS, int HOOK> // HOOK 1: the filter's load / store programs walked per sample; 2: the load program is one rotator on a complex stream (fir_band_hooks.hpp: BdRotor)
__global__ __launch_bounds__(256) void fir_decim_bf16x3_kernel(const float* __restrict__ x, const float* __restrict__ hist /*hist[h] = x[-Kh + h]*/, int Kh, const u32x4_b* __restrict__ afrag,
                                                                float* __restrict__ y, long n_out, long n_in, int D, int Hb /*a multiple of 4: samples in front of a block*/,
                                                                float* __restrict__ new_hist, BdHooks hk) {
    extern __shared__ __attribute__((aligned(16))) unsigned short bpl[]; // [3][PL]
    __shared__ float jst[2][8]; // the judge's sums, by segment parity
    const int NS = 16 * D * 63 + 32 * KS, PL = NS + 8; // staged samples per segment: the last block's window ends 16 D 63 + 32 KS
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
    u32x4_b   a[3][KS];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[p][ks] = afrag[(p * KS + ks) * 64 + lane];
    float4 nxt[kBdMaxNL4];
    auto   load_next = [&](long in0) { // in0 = first staged stream position (>= 0 here)
        const long   nrec = n_in - in0 < (long)NS ? n_in - in0 : (long)NS;
        const rsrc_t r    = make_rsrc(x + in0, (unsigned)(nrec > 0 ? nrec * 4 : 0));
#pragma unroll
        for (int u = 0; u < kBdMaxNL4; ++u) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, 256 * u * 16, 0);
            nxt[u]       = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto put4 = [&](int q, float4 v) {
        unsigned h0, m0, l0, h1, m1, l1;
        bf_split2(v.x, v.y, h0, m0, l0);
        bf_split2(v.z, v.w, h1, m1, l1);
        *reinterpret_cast<uint2*>(bpl + 4 * q)          = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(bpl + PL + 4 * q)     = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(bpl + 2 * PL + 4 * q) = make_uint2(l0, l1);
    };
    const long nseg = (n_out + kBdSegOut - 1) / kBdSegOut, sfirst = (long)blockIdx.x * kBfSegPerWg, slast = sfirst + kBfSegPerWg < nseg ? sfirst + kBfSegPerWg : nseg;
    auto in_start = [&](long sg) { return sg * kBdSegOut * D - Hb; };
    if (sfirst < slast && in_start(sfirst) >= 0) load_next(in_start(sfirst));
    for (long sg = sfirst; sg < slast; ++sg) {
        const long in0 = in_start(sg);
        float      pxl = 0.f;
        if (in0 >= 0) {
            [[maybe_unused]] BdRotor rot;
            if constexpr (HOOK == 2) rot = bd_rotor_start(hk.pre, in0, tid);
#pragma unroll
            for (int u = 0; u < kBdMaxNL4; ++u) {
                const int q = tid + 256 * u;
                if (q < NS / 4) {
                    if constexpr (HOOK == 2) nxt[u] = bd_rotor_next(nxt[u], rot);
                    if constexpr (HOOK == 1) { if (hk.pre.n_ops > 0) nxt[u] = bd_hook4(nxt[u], hk.pre, hk.cplx, in0 + 4L * q); }
                    put4(q, nxt[u]);
                    pxl = fmaf(nxt[u].x, nxt[u].x, fmaf(nxt[u].y, nxt[u].y, fmaf(nxt[u].z, nxt[u].z, fmaf(nxt[u].w, nxt[u].w, pxl))));
                }
            }
        } else { // the first segment of the span reads the carried history in front of x
            for (int q = tid; q < NS / 4; q += 256) {
                const float4 v = bd_stage_slow<HOOK != 0>(x, hist, Kh, n_in, in0 + 4L * q, hk);
                put4(q, v);
                pxl = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, pxl))));
            }
        }
        if (hk.flags != nullptr) {
            pxl = hf_wave_sum(pxl);
            if (lane == 0) jst[sg & 1][wave] = pxl;
        }
        __syncthreads();
        if (sg + 1 < slast) load_next(in_start(sg + 1)); // (>= 0: sg + 1 >= 1)
        f32x4_b c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0; // one accumulator per term: hh, hm, mh, hl, lh, mm
        const int s0 = 16 * D * (16 * wave + col) + 8 * kq;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const unsigned short* q0 = bpl + s0 + 32 * ks;
            const bf16x8 bh = *reinterpret_cast<const bf16x8*>(q0), bm = *reinterpret_cast<const bf16x8*>(q0 + PL), bl = *reinterpret_cast<const bf16x8*>(q0 + 2 * PL);
            const bf16x8 ah = __builtin_bit_cast(bf16x8, a[0][ks]), am = __builtin_bit_cast(bf16x8, a[1][ks]), al = __builtin_bit_cast(bf16x8, a[2][ks]);
            c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c0, 0, 0, 0);
            c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c1, 0, 0, 0);
            c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c2, 0, 0, 0);
#ifndef GR4_T_BD_THREE // timing only (wrong results): what three products instead of six would give this kernel
            c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c3, 0, 0, 0);
            c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c4, 0, 0, 0);
            c5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c5, 0, 0, 0);
#endif
        }
        const long o = sg * kBdSegOut + 16L * (16 * wave + col) + 4 * kq; // D[row = 4 kq + r][col]
        float      v[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) v[r] = (c0[r] + (c1[r] + c2[r])) + ((c3[r] + c4[r]) + c5[r]);
        float pyl = 0.f; // (the filter's output, in front of the store program)
#pragma unroll
        for (int r = 0; r < 4; ++r) pyl = o + r < n_out ? fmaf(v[r], v[r], pyl) : pyl;
        if constexpr (HOOK == 1) {
            if (hk.post.n_ops > 0) { const float4 w = bd_hook4(make_float4(v[0], v[1], v[2], v[3]), hk.post, hk.cplx, o); v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w; }
        }
        if (o + 3 < n_out) *reinterpret_cast<float4*>(y + o) = make_float4(v[0], v[1], v[2], v[3]);
        else
            for (int r = 0; r < 4; ++r)
                if (o + r < n_out) y[o + r] = v[r];
        if (hk.flags != nullptr) {
            pyl = hf_wave_sum(pyl);
            if (lane == 0) jst[sg & 1][4 + wave] = sg * kBdSegOut + 256L * wave < n_out ? pyl : __builtin_inff(); // (a wave past the end of the span: nothing to judge)
        }
        __syncthreads();
        if (hk.flags != nullptr && tid == 0) bd_judge(jst[sg & 1], hk, sg, D);
    }
    if (new_hist != nullptr && blockIdx.x == 0) bd_new_hist<HOOK != 0>(x, hist, Kh, n_in, new_hist, tid, hk);
}

// The same with LONG windows (taps - 1 + 15 D + 1 <= 1152 samples: BASELINE configs[2]'s decimate-by-8 1024-tap filter, decimation 10 .. 32 with K ~ 32 D): the four
// waves of a workgroup split the K-steps -- each holds the fragments of its quarter (<= 9 steps) -- and their partial tiles are summed through LDS.  Two tiles
// (512 outputs, 512 D inputs + the window) per segment, twelve accumulators per wave.
constexpr int kBsTiles = 2, kBsSegOut = 256 * kBsTiles, kBsMaxNL4 = 20;
template <int KSW, int NL4, int HOOK> // K-steps of 32 per wave (the window is 128 KSW samples); float4 loads a lane holds for the next segment
__global__ __launch_bounds__(256) void fir_decim_bf16x3_splitk_kernel(const float* __restrict__ x, const float* __restrict__ hist /*hist[h] = x[-Kh + h]*/, int Kh, const u32x4_b* __restrict__ afrag /*[3][4 KSW][64]*/,
                                                                       float* __restrict__ y, long n_out, long n_in, int D, int Hb, float* __restrict__ new_hist, int spw /*segments per workgroup*/,
                                                                       BdHooks hk) {
    extern __shared__ __attribute__((aligned(16))) unsigned short bpl[]; // [3][PL] bf16 planes, then the partial tiles [4 waves][kBsTiles][64 lanes][4] floats
    __shared__ float jst[2][8]; // the judge's sums, by segment parity
    constexpr int KS = 4 * KSW;
    const int NS = 16 * D * (16 * kBsTiles - 1) + 32 * KS, PL = NS + 8;
    float*    part = reinterpret_cast<float*>(bpl + 3 * PL + (3 * PL & 1));
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
    u32x4_b   a[3][KSW];
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
        for (int ks = 0; ks < KSW; ++ks) a[p][ks] = afrag[(p * KS + wave * KSW + ks) * 64 + lane];
    float4 nxt[NL4];
    auto   load_next = [&](long in0) {
        const long   nrec = n_in - in0 < (long)NS ? n_in - in0 : (long)NS;
        const rsrc_t r    = make_rsrc(x + in0, (unsigned)(nrec > 0 ? nrec * 4 : 0));
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, 256 * u * 16, 0);
            nxt[u]       = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto put4 = [&](int q, float4 v) {
        unsigned h0, m0, l0, h1, m1, l1;
        bf_split2(v.x, v.y, h0, m0, l0);
        bf_split2(v.z, v.w, h1, m1, l1);
        *reinterpret_cast<uint2*>(bpl + 4 * q)          = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(bpl + PL + 4 * q)     = make_uint2(m0, m1);
        *reinterpret_cast<uint2*>(bpl + 2 * PL + 4 * q) = make_uint2(l0, l1);
    };
    const long nseg = (n_out + kBsSegOut - 1) / kBsSegOut, sfirst = (long)blockIdx.x * spw, slast = sfirst + spw < nseg ? sfirst + spw : nseg;
    auto in_start = [&](long sg) { return sg * kBsSegOut * D - Hb; };
    if (sfirst < slast && in_start(sfirst) >= 0) load_next(in_start(sfirst));
    for (long sg = sfirst; sg < slast; ++sg) {
        const long in0 = in_start(sg);
        float      pxl = 0.f, pyl = 0.f;
        if (in0 >= 0) {
            [[maybe_unused]] BdRotor rot;
            if constexpr (HOOK == 2) rot = bd_rotor_start(hk.pre, in0, tid);
#pragma unroll
            for (int u = 0; u < NL4; ++u) {
                const int q = tid + 256 * u;
                if (q < NS / 4) {
                    if constexpr (HOOK == 2) nxt[u] = bd_rotor_next(nxt[u], rot);
                    if constexpr (HOOK == 1) { if (hk.pre.n_ops > 0) nxt[u] = bd_hook4(nxt[u], hk.pre, hk.cplx, in0 + 4L * q); }
                    put4(q, nxt[u]);
                    pxl = fmaf(nxt[u].x, nxt[u].x, fmaf(nxt[u].y, nxt[u].y, fmaf(nxt[u].z, nxt[u].z, fmaf(nxt[u].w, nxt[u].w, pxl))));
                }
            }
        } else {
            for (int q = tid; q < NS / 4; q += 256) {
                const float4 v = bd_stage_slow<HOOK != 0>(x, hist, Kh, n_in, in0 + 4L * q, hk);
                put4(q, v);
                pxl = fmaf(v.x, v.x, fmaf(v.y, v.y, fmaf(v.z, v.z, fmaf(v.w, v.w, pxl))));
            }
        }
        if (hk.flags != nullptr) {
            pxl = hf_wave_sum(pxl);
            if (lane == 0) jst[sg & 1][wave] = pxl;
        }
        __syncthreads();
        if (sg + 1 < slast) load_next(in_start(sg + 1));
#pragma unroll
        for (int tl = 0; tl < kBsTiles; ++tl) {
            f32x4_b   c0 = {0.f, 0.f, 0.f, 0.f}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0;
            const int s0 = 16 * D * (16 * tl + col) + 32 * (wave * KSW) + 8 * kq; // this wave's quarter of the window
#pragma unroll
            for (int ks = 0; ks < KSW; ++ks) {
                const unsigned short* q0 = bpl + s0 + 32 * ks;
                const bf16x8 bh = *reinterpret_cast<const bf16x8*>(q0), bm = *reinterpret_cast<const bf16x8*>(q0 + PL), bl = *reinterpret_cast<const bf16x8*>(q0 + 2 * PL);
                const bf16x8 ah = __builtin_bit_cast(bf16x8, a[0][ks]), am = __builtin_bit_cast(bf16x8, a[1][ks]), al = __builtin_bit_cast(bf16x8, a[2][ks]);
                c0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c1, 0, 0, 0);
                c2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c2, 0, 0, 0);
                c3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, c3, 0, 0, 0);
                c4 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, c4, 0, 0, 0);
                c5 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, c5, 0, 0, 0);
            }
            float4 v;
            v.x = (c0[0] + (c1[0] + c2[0])) + ((c3[0] + c4[0]) + c5[0]);
            v.y = (c0[1] + (c1[1] + c2[1])) + ((c3[1] + c4[1]) + c5[1]);
            v.z = (c0[2] + (c1[2] + c2[2])) + ((c3[2] + c4[2]) + c5[2]);
            v.w = (c0[3] + (c1[3] + c2[3])) + ((c3[3] + c4[3]) + c5[3]);
            *reinterpret_cast<float4*>(part + ((wave * kBsTiles + tl) * 64 + lane) * 4) = v;
        }
        __syncthreads();
        // D[row = 4 kq + r][col] of tile tl: output 256 tl + 16 col + 4 kq + r of the segment; thread t sums the four waves' partial sums of outputs t and t + 256
#pragma unroll
        for (int tl = 0; tl < kBsTiles; ++tl) {
            const int  o = tid, ln = (o >> 4) + 16 * ((o & 15) >> 2), r = o & 3;
            const long m = sg * kBsSegOut + 256 * tl + o;
            float      val = (part[((0 * kBsTiles + tl) * 64 + ln) * 4 + r] + part[((1 * kBsTiles + tl) * 64 + ln) * 4 + r]) +
                             (part[((2 * kBsTiles + tl) * 64 + ln) * 4 + r] + part[((3 * kBsTiles + tl) * 64 + ln) * 4 + r]);
            if (m < n_out) pyl = fmaf(val, val, pyl); // (the filter's output, in front of the store program)
            if constexpr (HOOK == 1) {
                if (hk.post.n_ops > 0) {
                    if (hk.cplx) { // lanes 2 i and 2 i + 1 hold one complex output: both evaluate the program on it, each keeps its component
                        const float other = __shfl_xor(val, 1);
                        const float2 w    = ewise_hook1<float2>((tid & 1) ? make_float2(other, val) : make_float2(val, other), hk.post, m >> 1);
                        val               = (tid & 1) ? w.y : w.x;
                    } else val = ewise_hook1<float>(val, hk.post, m);
                }
            }
            if (m < n_out) y[m] = val;
        }
        if (hk.flags != nullptr) { // (a wave's 64 threads hold outputs 64 w .. 64 w + 63 of both tiles)
            pyl = hf_wave_sum(pyl);
            if (lane == 0) jst[sg & 1][4 + wave] = sg * kBsSegOut + 64L * wave < n_out ? pyl : __builtin_inff();
        }
        __syncthreads(); // (the partial tiles and the planes are reused by the next segment)
        if (hk.flags != nullptr && tid == 0) bd_judge(jst[sg & 1], hk, sg, D);
    }
    if (new_hist != nullptr && blockIdx.x == 0) bd_new_hist<HOOK != 0>(x, hist, Kh, n_in, new_hist, tid, hk);
}

static unsigned short host_bf_rne(float f) {
    unsigned u;
    std::memcpy(&u, &f, 4);
    u += 0x7FFFu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
static float host_bf_to_f(unsigned short h) {
    const unsigned u = (unsigned)h << 16;
    float          f;
    std::memcpy(&f, &u, 4);
    return f;
}

// fragment tables [3][KS][64][8] of bf16: plane p, K-step ks, lane l, element t = tap-plane value b_p[Hb + (l & 15) - (32 ks + 8 (l >> 4) + t)]
// (nch channels with their own taps [nch][ntaps]: the tables follow each other)
void fir_bf16_make_afrag(const float* taps_all, size_t ntaps, int* KS_out, std::vector<unsigned short>* af, size_t nch, int force_ks) {
    const int KS = force_ks ? force_ks : std::max(3, (int)((ntaps - 1 + 16 + 31) / 32)), Hb = 32 * KS - 16; // the smallest window of 32 KS samples with Hb = 32 KS - 16 >= taps - 1 (KS = 3 .. 9)
    af->assign((size_t)nch * 3 * KS * 64 * 8, 0);
    for (size_t c = 0; c < nch; ++c) {
    const float* taps = taps_all + c * ntaps;
    std::vector<unsigned short> pl[3];
    for (auto& v : pl) v.assign(ntaps, 0);
    for (size_t k = 0; k < ntaps; ++k) {
        const float          b = taps[k];
        const unsigned short h = host_bf_rne(b);
        const float          r1 = b - host_bf_to_f(h);
        const unsigned short m = host_bf_rne(r1);
        const float          r2 = r1 - host_bf_to_f(m);
        pl[0][k] = h;
        pl[1][k] = m;
        pl[2][k] = host_bf_rne(r2);
    }
    for (int p = 0; p < 3; ++p)
        for (int ks = 0; ks < KS; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int t = 0; t < 8; ++t) {
                    const int k = Hb + (l & 15) - (32 * ks + 8 * (l >> 4) + t);
                    if (k >= 0 && (size_t)k < ntaps) (*af)[((((size_t)c * 3 + p) * KS + ks) * 64 + l) * 8 + t] = pl[p][k];
                }
    }
    *KS_out = KS;
}

// y[i] = sum_k b[k] x[i - k], i < n; hist = the Kh samples in front of x; x and y 16-byte aligned
int fir_bf16_launch(int KS, const float* x, long n, const float* hist, int Kh, const void* afrag, float* y, hipStream_t st, float* new_hist, long in_stride, long out_stride, unsigned nch, int delay,
                    int accum) {
    const auto af = static_cast<const u32x4_b*>(afrag);
    if ((KS == 8 || KS == 9) && KS >= kBfSharedMinKS && 32 * KS - 16 + delay <= kBfSeg) { // the wide windows: shared fragment stream, double-buffered planes (fir_mfma_bf16x3_shared_kernel)
        const long nseg = ceil_div(n, (long)kBfSeg);
        const int  spw  = (int)std::min<long>(std::max<long>(nseg * (long)nch / GR4_BF16_TARGET_WGS, 1), GR4_BF16_MAX_SPW); // segments per workgroup: the prologue (tap fragments, first staging) once per run
        const dim3 grid((unsigned)ceil_div(nseg, (long)spw), nch);
        if (KS == 8) hipLaunchKernelGGL(fir_mfma_bf16x3_shared_kernel<8>, grid, dim3(256), 0, st, x, hist, Kh, af, y, n, new_hist, in_stride, out_stride, delay, accum, spw);
        else hipLaunchKernelGGL(fir_mfma_bf16x3_shared_kernel<9>, grid, dim3(256), 0, st, x, hist, Kh, af, y, n, new_hist, in_stride, out_stride, delay, accum, spw);
        GR4_LAUNCH_CHECK();
        return GR4HIP_OK;
    }
    const dim3 grid((unsigned)ceil_div(ceil_div(n, (long)kBfSeg), (long)kBfSegPerWg), nch);
#define GR4_BF_CASE(K) case K: hipLaunchKernelGGL(fir_mfma_bf16x3_kernel<K>, grid, dim3(256), 0, st, x, hist, Kh, af, y, n, new_hist, in_stride, out_stride, delay, accum); break
    switch (KS) {
        GR4_BF_CASE(3);
        GR4_BF_CASE(4);
        GR4_BF_CASE(5);
        GR4_BF_CASE(6);
        GR4_BF_CASE(7);
        GR4_BF_CASE(8);
        GR4_BF_CASE(9);
    default: return GR4HIP_UNSUPPORTED;
    }
#undef GR4_BF_CASE
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

// the same on complex samples; hist = the Kh complex samples in front of x; x and y 16-byte aligned
int fir_bf16_c32_launch(int KS, const float* x, long n, const float* hist, int Kh, const void* afrag, float* y, hipStream_t st, float* new_hist) {
    const dim3 grid((unsigned)ceil_div(ceil_div(n, (long)kBfSegC), (long)kBfSegPerWg));
    const auto af = static_cast<const u32x4_b*>(afrag);
    const auto xc = reinterpret_cast<const float2*>(x), hc = reinterpret_cast<const float2*>(hist);
    const auto yc = reinterpret_cast<float2*>(y), nh = reinterpret_cast<float2*>(new_hist);
#define GR4_BFC_CASE(K) case K: hipLaunchKernelGGL(fir_mfma_bf16x3_c32_kernel<K>, grid, dim3(256), 0, st, xc, hc, Kh, af, yc, n, nh); break
    switch (KS) {
        GR4_BFC_CASE(3);
        GR4_BFC_CASE(4);
        GR4_BFC_CASE(5);
        GR4_BFC_CASE(6);
        GR4_BFC_CASE(7);
        GR4_BFC_CASE(8);
        GR4_BFC_CASE(9);
    default: return GR4HIP_UNSUPPORTED;
    }
#undef GR4_BFC_CASE
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

// band-form fragments of a decimator: [3][KS][64][8], element t of lane l at K-step ks = tap-plane value b_p[Hb + (l & 15) D - (32 ks + 8 (l >> 4) + t)]
// Returns KS = 0 when the window Hb + 15 D + 1 does not fit 1152 samples (KS <= 9: one wave holds all fragments; 12 .. 36: split over the four waves).
// cplx: complex<float> samples x real taps on the SAME kernels, the interleaved stream read as 2 n floats: float output o = 2 m + c (component c of output m) is
// sum_k b[k] xf[2 (m D - k) + c], i.e. row j of a tile (16 float outputs = 8 complex ones, 16 D floats of input further on per tile, exactly the float case) sits
// s_j = 2 D (j >> 1) + (j & 1) floats into the window and sees tap k at window position Hb + s_j - 2 k: A[j][u] = b[(Hb + s_j - u) / 2] where that is even, else 0.
// Half of the matrix pipe's work multiplies zeros; the pipe has that to spare (the kernels are bound by the stream), the VALU polyphase kernel it replaces is not close.
void fir_decim_bf16_make_afrag(const float* taps, size_t ntaps, size_t D, int* KS_out, int* Hb_out, std::vector<unsigned short>* af, bool cplx) {
    const int Hb = (int)(((cplx ? 2 : 1) * (ntaps - 1) + 3) / 4 * 4);
    int       KS = ((Hb + (cplx ? 14 * (int)D + 1 : 15 * (int)D) + 1 + 31) / 32 * 32) / 32;
    *KS_out = 0;
    *Hb_out = Hb;
    if (KS > 9) KS = (KS + 3) / 4 * 4; // split over the four waves (fir_decim_bf16x3_splitk_kernel): a multiple of 4, <= 36
    if (KS < 1 || KS > 36) return;
    std::vector<unsigned short> pl[3];
    for (auto& v : pl) v.assign(ntaps, 0);
    for (size_t k = 0; k < ntaps; ++k) {
        const float          b = taps[k];
        const unsigned short h = host_bf_rne(b);
        const float          r1 = b - host_bf_to_f(h);
        const unsigned short m = host_bf_rne(r1);
        pl[0][k] = h;
        pl[1][k] = m;
        pl[2][k] = host_bf_rne(r1 - host_bf_to_f(m));
    }
    af->assign((size_t)3 * KS * 64 * 8, 0);
    for (int p = 0; p < 3; ++p)
        for (int ks = 0; ks < KS; ++ks)
            for (int l = 0; l < 64; ++l)
                for (int t = 0; t < 8; ++t) {
                    const int j = l & 15;
                    long      k = (long)Hb + (cplx ? 2L * (long)D * (j >> 1) + (j & 1) : (long)j * (long)D) - (32 * ks + 8 * (l >> 4) + t);
                    if (cplx) k = (k & 1) ? -1 : k / 2;
                    if (k >= 0 && (size_t)k < ntaps) (*af)[(((size_t)p * KS + ks) * 64 + l) * 8 + t] = pl[p][(size_t)k];
                }
    *KS_out = KS;
}

// y[m] = sum_k b[k] x[m D - k], m < n_out; hist[h] = x[-Kh + h]; x and y 16-byte aligned
// pre / post: programs for the samples on their way in / the outputs on their way out (null: none); cplx: the stream is complex<float> read as floats (the programs'
// positions are sample indices)
// flags (optional): one byte per segment -- *seg_out outputs (floats, as n_out counts): 512 for the split-K kernel (KS > 9), else 1024 -- the segments whose
// output power is below gthr x their input power: fir_exact_launch evaluates them again behind this launch
int fir_decim_bf16_launch(int KS, int D, int Hb, const float* x, long n_in, const float* hist, int Kh, const void* afrag, float* y, long n_out, hipStream_t st, float* new_hist,
                          const EwiseHook* pre, const EwiseHook* post, bool cplx, unsigned char* flags, float gthr, int* seg_out) {
    BdHooks hk;
    if (pre) hk.pre = *pre;
    if (post) hk.post = *post;
    hk.cplx = cplx ? 1 : 0;
    const bool hooked = hk.pre.n_ops > 0 || hk.post.n_ops > 0;
    if (flags != nullptr && gthr > 0.f) { hk.flags = flags; hk.gthr = gthr; }
    if (seg_out) *seg_out = KS > 9 ? kBsSegOut : kBdSegOut;
    if (hooked && (Kh % 4) != 0) return GR4HIP_UNSUPPORTED;
    const bool ddc = cplx && hk.pre.rotor_only && hk.pre.n_ops == 1 && hk.post.n_ops == 0; // rotator -> decimator: the phase stepped in integers (BdRotor)
    if (KS > 9) { // long window: the waves split the K-steps
        const int    NS   = 16 * D * (16 * kBsTiles - 1) + 32 * KS, PL = NS + 8;
        const size_t lds  = (size_t)(3 * PL + (3 * PL & 1)) * sizeof(unsigned short) + (size_t)4 * kBsTiles * 64 * 4 * sizeof(float);
        if (lds > 72 * 1024 || (NS / 4 + 255) / 256 > kBsMaxNL4 || KS % 4) return GR4HIP_UNSUPPORTED;
        const bool small = (NS / 4 + 255) / 256 <= 6; // (fewer prefetch registers: one more wave per SIMD)
        const long nseg = ceil_div(n_out, (long)kBsSegOut);
        const int  spw  = (int)std::min<long>(std::max<long>(nseg / 2048, 1), 8);
        const dim3 grid((unsigned)ceil_div(nseg, (long)spw));
        const auto af   = static_cast<const u32x4_b*>(afrag);
#define GR4_BS_CASE(K)                                                                                                                     \
    case K: {                                                                                                                              \
        auto kern = ddc    ? (small ? fir_decim_bf16x3_splitk_kernel<K, 6, 2> : fir_decim_bf16x3_splitk_kernel<K, kBsMaxNL4, 2>)           \
                    : hooked ? (small ? fir_decim_bf16x3_splitk_kernel<K, 6, 1> : fir_decim_bf16x3_splitk_kernel<K, kBsMaxNL4, 1>)         \
                             : (small ? fir_decim_bf16x3_splitk_kernel<K, 6, 0> : fir_decim_bf16x3_splitk_kernel<K, kBsMaxNL4, 0>);        \
        if (lds > 48 * 1024) GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, hist, Kh, af, y, n_out, n_in, D, Hb, new_hist, spw, hk);                     \
    } break
        switch (KS / 4) {
            GR4_BS_CASE(3);
            GR4_BS_CASE(4);
            GR4_BS_CASE(5);
            GR4_BS_CASE(6);
            GR4_BS_CASE(7);
            GR4_BS_CASE(8);
            GR4_BS_CASE(9);
        default: return GR4HIP_UNSUPPORTED;
        }
#undef GR4_BS_CASE
        GR4_LAUNCH_CHECK();
        return GR4HIP_OK;
    }
    const int    NS  = 16 * D * 63 + 32 * KS;
    const size_t lds = (size_t)3 * (NS + 8) * sizeof(unsigned short);
    if (lds > 64 * 1024 || (NS / 4 + 255) / 256 > kBdMaxNL4) return GR4HIP_UNSUPPORTED;
    const dim3 grid((unsigned)ceil_div(ceil_div(n_out, (long)kBdSegOut), (long)kBfSegPerWg));
    const auto af = static_cast<const u32x4_b*>(afrag);
#define GR4_BD_CASE(K)                                                                                                                     \
    case K: {                                                                                                                              \
        auto kern = ddc ? fir_decim_bf16x3_kernel<K, 2> : (hooked ? fir_decim_bf16x3_kernel<K, 1> : fir_decim_bf16x3_kernel<K, 0>);        \
        if (lds > 48 * 1024) GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, hist, Kh, af, y, n_out, n_in, D, Hb, new_hist, hk);                          \
    } break
    switch (KS) {
        GR4_BD_CASE(1);
        GR4_BD_CASE(2);
        GR4_BD_CASE(3);
        GR4_BD_CASE(4);
        GR4_BD_CASE(5);
        GR4_BD_CASE(6);
        GR4_BD_CASE(7);
        GR4_BD_CASE(8);
        GR4_BD_CASE(9);
    default: return GR4HIP_UNSUPPORTED;
    }
#undef GR4_BD_CASE
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

} // namespace gr4
