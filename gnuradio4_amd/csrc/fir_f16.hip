// fir_f16.hip -- fir_filter<float> (33 .. 256 taps, slices of longer filters, the batched many-channel FIR) on the f16 matrix pipe: two-term splits under a
// per-segment block exponent, three products per tap instead of fir_bf16.hip's six.
//
// fir_bf16.hip holds the package at its 1400 W cap with the shader clock at 1.73 - 1.77 GHz: six dense bf16 MFMAs per K-step are the price of float32 accuracy on a
// pipe whose terms carry 8 significant bits.  An f16 term carries 11: x = (x1 + x2 / 2^11) / s with x1 = f16(x s), x2 = f16((x s - x1) 2^11) is exact to 2^-22 |x|
// (typically 2^-24), and with the taps split the same way the three products x1 b1, x1 b2, x2 b1 carry everything above 2^-22 of a product -- the parity bar is
// 1e-5 = 2^-16.6.  Half the matrix-pipe work of the bf16 form, two planes instead of three in LDS.  (The fourth product x2 b2 was measured and dropped: 256 taps
// 526 -> 453 Gsamples/s for an error of 1.4e-4 instead of 1.8e-4 of the output under a rejected tone 50 dB above it -- the representation of x and b in 22 bits is
// what is left either way, and that regime is the guard's below.)
//
// What f16 does not have is float32's exponent range, so each segment of 4096 outputs (its 4096 + Hb staged samples) carries ONE block exponent: s is the power
// of two that puts the segment's largest magnitude in [2^14, 2^15); the residual plane is scaled by another 2^11, so its quantisation floor (f16 subnormals,
// 2^-24) sits 2^-49 below the segment's largest sample.  The statistics (largest magnitude, and the quietest group of four consecutive samples as the level of
// the ordinary samples) are taken from the registers the segment is prefetched into, one segment ahead of the split, through eight LDS words -- no extra barrier.
// A segment that holds a non-finite sample, or whose largest sample is more than 2^28 above that ordinary level (a glitch of 1e30 beside unit-power samples, a
// burst that ends inside the segment and leaves a floor 170 dB below it),
// is not given to the f16 pipe at all: the workgroup evaluates its 4096 outputs with float32 products on the f32 matrix pipe (slow_segment below) -- or, when the
// sample is not finite, as plain float32 sums one output at a time: the reference's +-Inf / NaN on exactly the ntaps outputs whose window holds it.
//
// The kernel also judges its own accuracy.  The error of this form is relative to the PRODUCTS (~1.3e-7 rms of sqrt(sum b^2) x rms(x)), like float32's own rounding
// but four times larger, so it only shows against the OUTPUT when the filter removes nearly everything it is given.  Every segment's output power is compared
// with its input power -- P_y as sixteen times the power of the QUIETEST of the segment's sixteen output columns (256 outputs each), so that a start-up transient or
// the edge of a burst somewhere in the segment does not hide that the rest of it is all rejection: P_y < 2^-12 (sum b^2) P_x -- more than 36 dB of the staged power
// rejected beyond what white noise would lose -- and the segment is evaluated again at the end of the workgroup's run of segments (`guard`; off for the slices of long
// filters, whose launches see partial sums) with THREE f16 terms per factor -- 33 bits, the float32 values themselves; the six products of order <= 2, each exact to 2^-33,
// in the same float32 accumulators: float32 products at twice the first evaluation's matrix-pipe time (redo_segment; the f32 matrix pipe would take five times).  So a
// stream whose rejected part is far above what passes costs three evaluations' worth on exactly the segments where that is so (256 taps: 190 instead of 515 Gsamples/s
// when EVERY segment is rejected -- after two rejections in a row a workgroup skips the first evaluation but for every eighth segment), and has float32 products there: measured 3.2e-5 .. 3.6e-5 under a tone 50 dB above the output, the f32 kernels' 3.4e-5 .. 4.0e-5, the
// reference's sequential float32 sum 6.8e-5 .. 7.3e-5 (include/gr4hip.h, "PARITY CONTRACT").
//
// (Measured and dropped: the outputs through LDS as whole 1 KiB rows one segment later instead of 64-byte pieces straight from the accumulators -- 256 taps 499 -> 488,
// 64 taps 623 -> 626 Gsamples/s on one box: the store pattern is not what these launches wait for.)
//
// Same output-to-tile map, fragment stream, double-buffered planes and prefetch as fir_mfma_bf16x3_shared_kernel (fir_bf16.hip), for every window width KS = 3 .. 9.
#ifndef GR4_HF_LOAD_AUX // cache policy of the sample loads / result stores (2 = nt, streaming): developer builds override; profiles/r05_streaming_hints.txt
#define GR4_HF_LOAD_AUX 0
#endif
#ifndef GR4_HF_STORE_AUX
#define GR4_HF_STORE_AUX 0
#endif
#include "common.hpp"
#include "buffer_ops.hpp"
#include "fir_f16_common.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>

namespace gr4 {

#ifndef GR4_F16_WG_PER_CU
#define GR4_F16_WG_PER_CU 2
#endif
#ifndef GR4_F16_TARGET_WGS
#define GR4_F16_TARGET_WGS 512
#define GR4_F16_MAX_SPW 128
#endif
constexpr int kHfSeg      = 4096;

// channel block of the table fir_f16_make_afrag writes, in 16-bit units: [3 planes][KS][64 lanes][8] f16 fragments (the third plane: the judged segments' second evaluation), 8 units of header {float 1 / t, int ntaps, float 2^-12 sum b^2, -},
// 32 KS float taps (the float32 path's)
__host__ __device__ constexpr int hf_block_units(int KS) { return KS * 1536 + 8 + KS * 64; }

template <int KS> // K-steps of 32: window Kw = 32 KS, Hb = Kw - 16 samples in front of a 16-output block
__global__ __launch_bounds__(256, GR4_F16_WG_PER_CU) void fir_mfma_f16x2_kernel(const float* __restrict__ x0, const float* __restrict__ hist0 /*the Kh samples in front of x*/, int Kh,
                                                                                const unsigned short* __restrict__ blk0 /*[channels] blocks of hf_block_units(KS)*/, float* __restrict__ y0, long n,
                                                                                float* __restrict__ new_hist, long in_stride, long out_stride /*channel blockIdx.y*/,
                                                                                int delay /*this pass filters x delayed by `delay` samples ...*/, int accum /*... and adds to y (slices of long filters)*/,
                                                                                int seg_per_wg, int guard /*judge every segment's output / input power (see the header)*/,
                                                                                unsigned char* __restrict__ flags0 /*one byte per segment: what fir_exact_kernel evaluates again behind this launch*/, long flags_stride,
                                                                                float gthr_arg /*> 0: the guard's threshold instead of the table's (the last slice of a long filter judges the whole filter's sum)*/) {
    const float*          x     = x0 + (long)blockIdx.y * in_stride;
    const float*          hist  = hist0 + (long)blockIdx.y * Kh;
    const unsigned short* blk   = blk0 + (long)blockIdx.y * hf_block_units(KS);
    const u32x4_h*        afrag = reinterpret_cast<const u32x4_h*>(blk);
    const float           inv_t = *reinterpret_cast<const float*>(blk + KS * 1536);
    const float           gthr  = gthr_arg > 0.f ? gthr_arg : *reinterpret_cast<const float*>(blk + KS * 1536 + 4);
    float*                y     = y0 + (long)blockIdx.y * out_stride;
    unsigned char*        flags = flags0 + (long)blockIdx.y * flags_stride;
    constexpr int Kw = 32 * KS, Hb = Kw - 16, NS = kHfSeg + Hb; // staged samples per segment (a multiple of 16)
    constexpr int PL  = NS + 8 * (NS / 256) + 16;               // f16 elements per plane: one 16-byte chunk of padding per 256 samples (see P below)
    constexpr int NL4 = (NS / 4 + 255) / 256;                   // float4 loads a lane holds for the next segment
    constexpr int NM  = KS + 3;                                 // fragments of a wave's stream
    __shared__ __attribute__((aligned(16))) unsigned short pls[2][2 * PL];
    __shared__ __attribute__((aligned(16))) unsigned stat[2][12]; // per data segment parity: the four waves' largest magnitude bits, quietest non-zero groups of four, sums of squares
    __shared__ __attribute__((aligned(16))) float    ystat[2][4][16]; // per computed segment parity: the four waves' output powers per column of 256 outputs
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
    auto P = [](int s_) { return s_ + 8 * (s_ >> 8); };

    u32x4_h a[2][KS]; // A fragments of the two tap planes
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[p][ks] = afrag[(p * KS + ks) * 64 + lane];

    auto xs = [&](long i) -> float { return i >= 0 ? (i < n ? x[i] : 0.f) : (i >= -(long)Kh ? hist[Kh + i] : 0.f); }; // the stream: carried history in front of x, zeros past the end
    float4 nxa[NL4], nxb[NL4];
    auto   load_next = [&](float4 (&nxt)[NL4], long seg0) { // seg0 >= kHfSeg >= Hb + delay: nothing below 0; past the end of the span / of the segment the range check returns 0
        const long   i0   = seg0 - Hb - delay;
        const long   nrec = n - i0 < (long)NS ? n - i0 : (long)NS;
        const rsrc_t r    = make_rsrc(x + i0, (unsigned)(nrec > 0 ? nrec * 4 : 0));
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, 256 * u * 16, GR4_HF_LOAD_AUX);
            nxt[u]       = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto load_general = [&](float4 (&nxt)[NL4], long seg0) { // the first segment of the span: sample by sample
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const int q = tid + 256 * u;
            float     t[4] = {0.f, 0.f, 0.f, 0.f};
            if (q < NS / 4)
                for (int c = 0; c < 4; ++c) t[c] = xs(seg0 + 4L * q + c - Hb - delay);
            nxt[u] = make_float4(t[0], t[1], t[2], t[3]);
        }
    };
    // statistics of the registers of one staged segment -> stat[slot]; read back (after a barrier) by block_scale
    auto put_stats = [&](const float4 (&v)[NL4], int slot) {
        float mf = 0.f, px = 0.f; // largest magnitude (v_max3_f32 on |.|: a NaN is passed over, an Inf stays) and the power (NaN if a NaN is among the samples)
        unsigned mn = 0xffffffffu; // the quietest group of four consecutive values that is not all zeros: the level of the ordinary samples, local in TIME (a level that
                                   // drops in the middle of a segment is seen, not only an isolated outlier)
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const float m4 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[u].x), __builtin_fabsf(v[u].y)), __builtin_fmaxf(__builtin_fabsf(v[u].z), __builtin_fabsf(v[u].w)));
            mf = __builtin_fmaxf(mf, m4);
            mn = min(mn, __float_as_uint(m4) - 1u); // (0 - 1 wraps to the largest value: groups of zeros are passed over; one unit on the others is nothing against 2^28)
            px = fmaf(v[u].x, v[u].x, fmaf(v[u].y, v[u].y, fmaf(v[u].z, v[u].z, fmaf(v[u].w, v[u].w, px))));
        }
        unsigned mx = __float_as_uint(mf);
        mx = hf_wave_reduce_u32(mx, [](unsigned a_, unsigned b_) { return a_ > b_ ? a_ : b_; });
        mn = hf_wave_reduce_u32(mn, [](unsigned a_, unsigned b_) { return a_ < b_ ? a_ : b_; });
        px = hf_wave_sum(px);
        if (lane == 0) { stat[slot][wave] = mx; stat[slot][4 + wave] = mn; stat[slot][8 + wave] = __float_as_uint(px); }
    };
    auto put_ypower = [&](float py, int slot) { // (a lane's tiles all lie in its column `col`)
        py = hf_column_sum(py);
        if (lane < 16) ystat[slot][wave][lane] = py;
    };
    // -> scale s (its inverse in inv_s), and whether the segment must take the float32 path
    auto block_scale = [&](int slot, float& s, float& inv_s, float& px) -> int { // 0: the f16 pipe; 1: float32 products (the spread); 2: plain float32 sums (a non-finite sample)
        const uint4 m4 = *reinterpret_cast<const uint4*>(&stat[slot][0]), n4 = *reinterpret_cast<const uint4*>(&stat[slot][4]), p4 = *reinterpret_cast<const uint4*>(&stat[slot][8]);
        px = (__uint_as_float(p4.x) + __uint_as_float(p4.y)) + (__uint_as_float(p4.z) + __uint_as_float(p4.w));
        const unsigned mx = __builtin_amdgcn_readfirstlane(max(max(m4.x, m4.y), max(m4.z, m4.w))), mn = __builtin_amdgcn_readfirstlane(min(min(n4.x, n4.y), min(n4.z, n4.w)));
        const int  e  = (int)(mx >> 23), el = (int)(mn >> 23);
        const int  slow = (e == 255 || px != px) ? 2 : ((mn != 0xffffffffu && e - el > kHfMaxRange) ? 1 : 0); // (px = +Inf without an Inf sample: squares above 3.4e38 -- see px = -1 below: the guard
                                                                                                                  //  cannot judge such a segment and sends it to the float32 products)
        if (slow == 0 && mx != 0u && (e < 127 - 60 || e > 127 + 60)) px = -1.f; // the powers are sums of SQUARES: with the largest sample outside [2^-60, 2^60] they leave float32's range (0 or Inf)
                                                                                   // and the guard cannot judge -- judge() hands such a segment to the second evaluation as if it were rejected
        const int  ec = e < 15 ? 15 : (e > 254 ? 254 : e);
        s     = __uint_as_float((unsigned)(268 - ec) << 23); // largest magnitude -> [2^14, 2^15)
        inv_s = __uint_as_float((unsigned)(ec - 14) << 23);
        return slow;
    };
    auto put4e = [&](unsigned short* pl, int e, float4 v, float s) {
        unsigned h0, l0, h1, l1;
        hf_split2(v.x, v.y, s, h0, l0);
        hf_split2(v.z, v.w, s, h1, l1);
        *reinterpret_cast<uint2*>(pl + e)      = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(pl + PL + e) = make_uint2(l0, l1);
    };
    auto put_next = [&](unsigned short* pl, int u, const float4& v, float s) { // branch-free (it sits between MFMAs): lanes past the staged range write into the spare elements behind each plane
        const int q = tid + 256 * u;
        put4e(pl, (256 * (u + 1) <= NS / 4 || q < NS / 4) ? P(4 * q) : PL - 16 + 4 * (lane & 3), v, s);
    };
    const long nseg = (n + kHfSeg - 1) / kHfSeg, sfirst = (long)blockIdx.x * seg_per_wg, slast = sfirst + seg_per_wg < nseg ? sfirst + seg_per_wg : nseg;
    if (sfirst >= slast) return; // (whole workgroups only: no barrier is skipped by part of one)
    if (!accum && tid < slast - sfirst) flags[sfirst + tid] = 0; // (a later slice of a long filter adds its marks to the earlier ones'; the barriers below order this store before any mark)
    const int tb = (wave >> 1) + 8 * (wave & 1); // waves 0, 1: the even tiles from 0 / 8; waves 2, 3: the odd tiles from 1 / 9
    const int sb = 256 * col + 16 * tb + 8 * kq;
    // segments this kernel cannot vouch for are only MARKED (a byte per segment in `flags`): fir_exact_kernel (fir_exact.hip), launched behind this one, evaluates their
    // outputs again on the FP64 matrix pipe.  (Until round 5 the second evaluation sat in this kernel behind the run: three more code paths, and their 110 live registers
    // spilled around the f16 loop.)
    auto note = [&](long sg, int kind) { // 1: the spread of the segment's samples is beyond the block exponent; 2: a non-finite sample; 3: rejected by the guard
        if (tid == 0) flags[sg] = (unsigned char)(accum ? (flags[sg] | kind) : kind);
    };
    // the guard's verdict on segment sg (its output powers are in ystat[sg & 1], a barrier ago): rejected -> noted for the second evaluation.  After two rejections in a row the
    // first evaluation of the following segments is skipped (they are noted unseen) but for every eighth, which probes whether the stream has changed: a stream that is all
    // rejection costs the second evaluation and an eighth of the first, not both
    int  streak = 0;
    auto judge = [&](long sg, float px) {
        const float pc = (ystat[sg & 1][0][col] + ystat[sg & 1][1][col]) + (ystat[sg & 1][2][col] + ystat[sg & 1][3][col]); // this lane's column, over the four waves' tiles
        const float py = 16.f * hf_row_min(pc); // the QUIETEST of the sixteen columns, as a segment's worth: a start-up transient or the edge of a burst in one part of the
                                                // segment does not hide that the rest of it is all rejection
        const int rej = __builtin_amdgcn_readfirstlane((int)(px < 0.f || py < gthr * px)), known = __builtin_amdgcn_readfirstlane((int)(py < __builtin_inff())); // (Inf: nothing was judged)
        if (rej) note(sg, 3);
        if (known) streak = rej ? streak + 1 : 0;
    };
    float s_cur, inv_cur, px_cur, px_prev = 0.f;
    int   slow_cur;
    {
        if (sfirst > 0) load_next(nxa, sfirst * kHfSeg);
        else load_general(nxa, 0);
        put_stats(nxa, (int)(sfirst & 1));
        __syncthreads();
        slow_cur = block_scale((int)(sfirst & 1), s_cur, inv_cur, px_cur);
#pragma unroll
        for (int u = 0; u < NL4; ++u) put_next(pls[0], u, nxa[u], s_cur);
        load_next(nxa, (sfirst + 1) * kHfSeg); // (past the end of the span: empty loads)
        put_stats(nxa, (int)((sfirst + 1) & 1));
        __syncthreads();
    }
    // one segment: MFMAs on `pl`; `cur` (the segment behind it, requested a segment ago, statistics in stat[(sg + 1) & 1]) goes into `plo` meanwhile; the segment behind THAT
    // is requested into `oth` first and its statistics are left in stat[sg & 1] before the barrier
    auto segment = [&](long sg, const unsigned short* pl, unsigned short* plo, float4 (&cur)[NL4], float4 (&oth)[NL4]) {
        const long seg0 = sg * kHfSeg;
        float      s_nx, inv_nx, px_nx;
        const int  slow_nx = block_scale((int)((sg + 1) & 1), s_nx, inv_nx, px_nx);
        load_next(oth, seg0 + 2 * kHfSeg);
        if (guard && sg > sfirst) judge(sg - 1, px_prev);
        const bool skip = guard && streak >= 2 && (sg & 7) != 0;
        float py = 0.f;
        if (!slow_cur && !skip) {
            f32x4_h c[4], d[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = d[j] = f32x4_h{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const unsigned short* q  = pl + P(sb + 32 * m);
                const f16x8_h         b1 = *reinterpret_cast<const f16x8_h*>(q), b2 = *reinterpret_cast<const f16x8_h*>(q + PL);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int ks = m - j;
                    if (ks < 0 || ks >= KS) continue;
                    const f16x8_h a1 = __builtin_bit_cast(f16x8_h, a[0][ks]), a2 = __builtin_bit_cast(f16x8_h, a[1][ks]);
                    c[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c[j], 0, 0, 0);
                    d[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, d[j], 0, 0, 0);
                    d[j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, d[j], 0, 0, 0);
                }
#pragma unroll
                for (int u = 0; u < NL4; ++u) // the next segment's samples are split and written into the other buffer beside the MFMAs, spread over the stream
                    if (u * NM / NL4 == m) put_next(plo, u, cur[u], s_nx);
            }
            // D[row = 4 kq + r][col]: y[seg0 + 256 col + 16 t + 4 kq + r]; the small terms are added to the large ones last, then the two block scales come off (powers of two:
            // as ONE factor when their product is a normal float -- every stream but those within 2^6 of float32's limits)
            const int   ek  = (int)((__float_as_uint(inv_t) >> 23) & 255) + (int)((__float_as_uint(inv_cur) >> 23) & 255) - 254;
            const bool  one = ek > -120 && ek < 120;
            const float k1 = one ? inv_t * inv_cur : inv_t, k2 = one ? 1.f : inv_cur, kd = k1 * (1.f / 2048.f);
            const bool  whole = seg0 + kHfSeg <= n, full = whole && !accum; // whole segments leave through a buffer descriptor (no 64-bit address per store)
            const rsrc_t ry  = make_rsrc(y + seg0, full ? kHfSeg * 4u : 0u);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    v[r] = fmaf(d[j][r], kd, c[j][r] * k1);
                    if (!one) v[r] *= k2;
                }
                const int oo = 256 * col + 16 * (tb + 2 * j) + 4 * kq;
                // the guard's output power: the outputs of the span only (past its end the staged zeros make the filter ring: not what it passes); the slices of a long
                // filter: of the sums this launch leaves in y (the last slice's are the filter's outputs)
                if (full) {
                    py = fmaf(v[0], v[0], fmaf(v[1], v[1], fmaf(v[2], v[2], fmaf(v[3], v[3], py))));
                    const u32x4_h w = {__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(w, ry, oo * 4, 0, GR4_HF_STORE_AUX);
                } else {
                    const long o = seg0 + oo;
                    if (o + 3 < n) {
                        float4 w = make_float4(v[0], v[1], v[2], v[3]);
                        if (accum) { const float4 p = *reinterpret_cast<const float4*>(y + o); w.x += p.x; w.y += p.y; w.z += p.z; w.w += p.w; }
                        py = fmaf(w.x, w.x, fmaf(w.y, w.y, fmaf(w.z, w.z, fmaf(w.w, w.w, py))));
                        *reinterpret_cast<float4*>(y + o) = w;
                    } else {
                        for (int r = 0; r < 4; ++r)
                            if (o + r < n) {
                                const float w = (accum ? y[o + r] : 0.f) + v[r];
                                py   = fmaf(w, w, py);
                                y[o + r] = w;
                            }
                    }
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < NL4; ++u) put_next(plo, u, cur[u], s_nx);
            note(sg, slow_cur ? slow_cur : 3);
            py = __builtin_inff(); // (nothing to judge)
        }
        if (seg0 + 256L * col >= n) py = __builtin_inff(); // (a column past the end of the span)
        put_stats(oth, (int)(sg & 1));
        put_ypower(py, (int)(sg & 1));
        s_cur    = s_nx;
        inv_cur  = inv_nx;
        slow_cur = slow_nx;
        px_prev  = px_cur;
        px_cur   = px_nx;
        __syncthreads(); // the other buffer and the statistics are complete, and every wave is done with this buffer before the segment after the next overwrites it
    };
    for (long sg = sfirst; sg < slast; sg += 2) {
        segment(sg, pls[0], pls[1], nxa, nxb);
        if (sg + 1 < slast) segment(sg + 1, pls[1], pls[0], nxb, nxa);
    }
    if (guard) judge(slast - 1, px_prev);
    if (new_hist != nullptr && blockIdx.x == 0 && blockIdx.y == 0) { // the other half of the caller's ping-pong pair: nobody reads it in this launch
        for (int h = tid; h < Kh; h += 256) {
            const long i = n - Kh + h;
            new_hist[h]  = i >= 0 ? x[i] : hist[Kh + i];
        }
    }
}

// fir_filter<complex<float>> (real taps on interleaved {re, im} samples) on the same scheme: the two components are two real streams under the same taps and ONE block
// exponent -- four planes (re / im x the two terms), 2048 complex outputs per segment (16 columns of 128, eight tiles per column), a wave walks the fragment stream of
// its two tiles (tb, tb + 2) once per component: four accumulator pairs in flight as in the float kernel.  D leaves re-interleaved, 32 bytes per lane.  The guard, the
// float32 products for judged / outlier segments and the plain sums for non-finite samples are the float kernel's, on both components.  hist: the Kh complex samples in
// front of x.
constexpr int kHfSegC = 2048;

// SL (round 5): a SLICE of a longer filter, as in the float kernel -- this pass filters x delayed by `delay` samples and, with `accum`, adds to what is in y; only the last slice
// judges (the sums it leaves in y, against the whole filter's threshold gthr_arg).  A template switch so that the 33 .. 256-tap kernels stay as they are (252 .. 256 registers at KS = 9).
template <int KS, bool SL = false>
__global__ __launch_bounds__(256, GR4_F16_WG_PER_CU) void fir_mfma_f16x2_c32_kernel(const float2* __restrict__ x, const float2* __restrict__ hist /*the Kh samples in front of x*/, int Kh,
                                                                                    const unsigned short* __restrict__ blk /*one block of hf_block_units(KS)*/, float2* __restrict__ y, long n,
                                                                                    float2* __restrict__ new_hist, int seg_per_wg, int guard,
                                                                                    unsigned char* __restrict__ flags /*one byte per segment: what fir_exact_kernel evaluates again behind this launch*/,
                                                                                    int delay_arg, int accum_arg, float gthr_arg) {
    const int delay = SL ? delay_arg : 0;
    const bool accum = SL && accum_arg != 0;
    const u32x4_h* afrag = reinterpret_cast<const u32x4_h*>(blk);
    const float    inv_t = *reinterpret_cast<const float*>(blk + KS * 1536);
    const float    gthr  = (SL && gthr_arg > 0.f) ? gthr_arg : *reinterpret_cast<const float*>(blk + KS * 1536 + 4);
    const float    gthr_all = *reinterpret_cast<const float*>(blk + KS * 1536 + 6); // (0 unless a library-internal caller set it -- fir.hip, gr4hip_internal_fir_set_guard_ratio: a second, tighter
                                                                                       // threshold on the segment's WHOLE output power, a statistic that does not dip on narrow-band noise as the quietest column does)
    constexpr int Kw = 32 * KS, Hb = Kw - 16, NS = kHfSegC + Hb; // staged complex samples per segment (a multiple of 16)
    constexpr int PL  = NS + 8 * (NS / 128 + 1) + 16;            // f16 elements per plane: one 16-byte chunk of padding per 128 samples (the columns are 128 samples apart)
    constexpr int NL4 = (NS / 2 + 255) / 256;                    // float4 loads (two complex samples each) a lane holds for the next segment
    constexpr int NM  = KS + 1;                                  // fragments of a wave's stream (two tiles, one K-step apart)
    __shared__ __attribute__((aligned(16))) unsigned short pls[2][4 * PL]; // planes re1, re2, im1, im2
    __shared__ __attribute__((aligned(16))) unsigned stat[2][12];
    __shared__ __attribute__((aligned(16))) float    ystat[2][4][16]; // (per column of 128 outputs)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
    auto P = [](int s_) { return s_ + 8 * (s_ >> 7); };

    u32x4_h a[2][KS];
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a[p][ks] = afrag[(p * KS + ks) * 64 + lane];

    auto xs = [&](long i) -> float2 { return i >= 0 ? (i < n ? x[i] : make_float2(0.f, 0.f)) : (i >= -(long)Kh ? hist[Kh + i] : make_float2(0.f, 0.f)); };
    float4 nxa[NL4], nxb[NL4];
    auto   load_next = [&](float4 (&nxt)[NL4], long seg0) { // seg0 >= kHfSegC >= Hb + delay
        const long   i0   = seg0 - Hb - delay;
        const long   nrec = n - i0 < (long)NS ? n - i0 : (long)NS;
        const rsrc_t r    = make_rsrc(x + i0, (unsigned)(nrec > 0 ? nrec * 8 : 0));
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, 256 * u * 16, GR4_HF_LOAD_AUX);
            nxt[u]       = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto load_general = [&](float4 (&nxt)[NL4], long seg0) {
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const int q = tid + 256 * u;
            float2    t0 = make_float2(0.f, 0.f), t1 = t0;
            if (q < NS / 2) { t0 = xs(seg0 + 2L * q - Hb - delay); t1 = xs(seg0 + 2L * q + 1 - Hb - delay); }
            nxt[u] = make_float4(t0.x, t0.y, t1.x, t1.y);
        }
    };
    auto put_stats = [&](const float4 (&v)[NL4], int slot) {
        float mf = 0.f, px = 0.f; // largest magnitude (v_max3_f32 on |.|: a NaN is passed over, an Inf stays) and the power (NaN if a NaN is among the samples)
        unsigned mn = 0xffffffffu; // the quietest group of four consecutive values that is not all zeros: the level of the ordinary samples, local in TIME (a level that
                                   // drops in the middle of a segment is seen, not only an isolated outlier)
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const float m4 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(v[u].x), __builtin_fabsf(v[u].y)), __builtin_fmaxf(__builtin_fabsf(v[u].z), __builtin_fabsf(v[u].w)));
            mf = __builtin_fmaxf(mf, m4);
            mn = min(mn, __float_as_uint(m4) - 1u); // (0 - 1 wraps to the largest value: groups of zeros are passed over; one unit on the others is nothing against 2^28)
            px = fmaf(v[u].x, v[u].x, fmaf(v[u].y, v[u].y, fmaf(v[u].z, v[u].z, fmaf(v[u].w, v[u].w, px))));
        }
        unsigned mx = __float_as_uint(mf);
        mx = hf_wave_reduce_u32(mx, [](unsigned a_, unsigned b_) { return a_ > b_ ? a_ : b_; });
        mn = hf_wave_reduce_u32(mn, [](unsigned a_, unsigned b_) { return a_ < b_ ? a_ : b_; });
        px = hf_wave_sum(px);
        if (lane == 0) { stat[slot][wave] = mx; stat[slot][4 + wave] = mn; stat[slot][8 + wave] = __float_as_uint(px); }
    };
    auto put_ypower = [&](float py, int slot) { // (a lane's tiles all lie in its column `col`)
        py = hf_column_sum(py);
        if (lane < 16) ystat[slot][wave][lane] = py;
    };
    auto block_scale = [&](int slot, float& s, float& inv_s, float& px) -> int {
        const uint4 m4 = *reinterpret_cast<const uint4*>(&stat[slot][0]), n4 = *reinterpret_cast<const uint4*>(&stat[slot][4]), p4 = *reinterpret_cast<const uint4*>(&stat[slot][8]);
        px = (__uint_as_float(p4.x) + __uint_as_float(p4.y)) + (__uint_as_float(p4.z) + __uint_as_float(p4.w));
        const unsigned mx = __builtin_amdgcn_readfirstlane(max(max(m4.x, m4.y), max(m4.z, m4.w))), mn = __builtin_amdgcn_readfirstlane(min(min(n4.x, n4.y), min(n4.z, n4.w)));
        const int e = (int)(mx >> 23), el = (int)(mn >> 23);
        const int slow = (e == 255 || px != px) ? 2 : ((mn != 0xffffffffu && e - el > kHfMaxRange) ? 1 : 0);
        if (slow == 0 && mx != 0u && (e < 127 - 60 || e > 127 + 60)) px = -1.f; // the powers are sums of SQUARES: with the largest sample outside [2^-60, 2^60] they leave float32's range (0 or Inf)
                                                                                   // and the guard cannot judge -- judge() hands such a segment to the second evaluation as if it were rejected
        const int ec = e < 15 ? 15 : (e > 254 ? 254 : e);
        s     = __uint_as_float((unsigned)(268 - ec) << 23);
        inv_s = __uint_as_float((unsigned)(ec - 14) << 23);
        return slow;
    };
    // two complex samples {re0, im0, re1, im1} -> elements e, e + 1 of the four planes
    auto put2e = [&](unsigned short* pl, int e, float4 v, float s) {
        unsigned rh, rl, ih, il;
        hf_split2(v.x, v.z, s, rh, rl);
        hf_split2(v.y, v.w, s, ih, il);
        *reinterpret_cast<unsigned*>(pl + e)          = rh;
        *reinterpret_cast<unsigned*>(pl + PL + e)     = rl;
        *reinterpret_cast<unsigned*>(pl + 2 * PL + e) = ih;
        *reinterpret_cast<unsigned*>(pl + 3 * PL + e) = il;
    };
    auto put_next = [&](unsigned short* pl, int u, const float4& v, float s) { // branch-free: lanes past the staged range write into the spare elements behind each plane
        const int q = tid + 256 * u;
        put2e(pl, (256 * (u + 1) <= NS / 2 || q < NS / 2) ? P(2 * q) : PL - 16 + 2 * (lane & 7), v, s);
    };
    const int tb = (wave >> 1) + 4 * (wave & 1); // waves 0, 1: tiles {0, 2} / {4, 6}; waves 2, 3: tiles {1, 3} / {5, 7}
    const int sb = 128 * col + 16 * tb + 8 * kq;
    const long nseg = (n + kHfSegC - 1) / kHfSegC, sfirst = (long)blockIdx.x * seg_per_wg, slast = sfirst + seg_per_wg < nseg ? sfirst + seg_per_wg : nseg;
    if (sfirst >= slast) return;
    if (!accum && tid < slast - sfirst) flags[sfirst + tid] = 0; // (the barriers below order this store before any mark; a later slice adds its marks to the earlier ones')
    auto note = [&](long sg, int kind) { // 1: the spread is beyond the block exponent; 2: a non-finite sample; 3: rejected by the guard -> fir_exact_kernel behind this launch
        if (tid == 0) flags[sg] = (unsigned char)(accum ? (flags[sg] | kind) : kind);
    };
    int  streak = 0; // (see the float kernel)
    auto judge = [&](long sg, float px) {
        const float pc = (ystat[sg & 1][0][col] + ystat[sg & 1][1][col]) + (ystat[sg & 1][2][col] + ystat[sg & 1][3][col]); // this lane's column, over the four waves' tiles
        const float py = 16.f * hf_row_min(pc); // the QUIETEST of the sixteen columns, as a segment's worth: a start-up transient or the edge of a burst in one part of the
                                                // segment does not hide that the rest of it is all rejection
        const int rej = __builtin_amdgcn_readfirstlane((int)(px < 0.f || py < gthr * px || hf_row_sum(pc) < gthr_all * px)), known = __builtin_amdgcn_readfirstlane((int)(py < __builtin_inff())); // (Inf: nothing was judged)
        if (rej) note(sg, 3);
        if (known) streak = rej ? streak + 1 : 0;
    };
    float s_cur, inv_cur, px_cur, px_prev = 0.f;
    int   slow_cur;
    {
        if (sfirst > 0) load_next(nxa, sfirst * kHfSegC);
        else load_general(nxa, 0);
        put_stats(nxa, (int)(sfirst & 1));
        __syncthreads();
        slow_cur = block_scale((int)(sfirst & 1), s_cur, inv_cur, px_cur);
#pragma unroll
        for (int u = 0; u < NL4; ++u) put_next(pls[0], u, nxa[u], s_cur);
        load_next(nxa, (sfirst + 1) * kHfSegC);
        put_stats(nxa, (int)((sfirst + 1) & 1));
        __syncthreads();
    }
    auto segment = [&](long sg, const unsigned short* pl, unsigned short* plo, float4 (&cur)[NL4], float4 (&oth)[NL4]) {
        const long seg0 = sg * kHfSegC;
        float      s_nx, inv_nx, px_nx;
        const int  slow_nx = block_scale((int)((sg + 1) & 1), s_nx, inv_nx, px_nx);
        load_next(oth, seg0 + 2 * kHfSegC);
        if (guard && sg > sfirst) judge(sg - 1, px_prev);
        const bool skip = guard && streak >= 2 && (sg & 7) != 0;
        float py = 0.f;
        if (!slow_cur && !skip) {
            f32x4_h c[4], d[4]; // index 2 tile + component
#pragma unroll
            for (int j = 0; j < 4; ++j) c[j] = d[j] = f32x4_h{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int m = 0; m < NM; ++m) {
                const unsigned short* q = pl + P(sb + 32 * m);
                f16x8_h               b1[2], b2[2];
#pragma unroll
                for (int cp = 0; cp < 2; ++cp) {
                    b1[cp] = *reinterpret_cast<const f16x8_h*>(q + 2 * cp * PL);
                    b2[cp] = *reinterpret_cast<const f16x8_h*>(q + (2 * cp + 1) * PL);
                }
#pragma unroll
                for (int jj = 0; jj < 2; ++jj) {
                    const int ks = m - jj;
                    if (ks < 0 || ks >= KS) continue;
                    const f16x8_h a1 = __builtin_bit_cast(f16x8_h, a[0][ks]), a2 = __builtin_bit_cast(f16x8_h, a[1][ks]);
#pragma unroll
                    for (int cp = 0; cp < 2; ++cp) {
                        c[2 * jj + cp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1[cp], c[2 * jj + cp], 0, 0, 0);
                        d[2 * jj + cp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2[cp], d[2 * jj + cp], 0, 0, 0);
                        d[2 * jj + cp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1[cp], d[2 * jj + cp], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < NL4; ++u)
                    if (u * NM / NL4 == m) put_next(plo, u, cur[u], s_nx);
            }
            const int   ek  = (int)((__float_as_uint(inv_t) >> 23) & 255) + (int)((__float_as_uint(inv_cur) >> 23) & 255) - 254;
            const bool  one = ek > -120 && ek < 120;
            const float k1 = one ? inv_t * inv_cur : inv_t, k2 = one ? 1.f : inv_cur, kd = k1 * (1.f / 2048.f);
            const bool   full = seg0 + kHfSegC <= n;
            const rsrc_t ry   = make_rsrc(y + seg0, full ? kHfSegC * 8u : 0u);
#pragma unroll
            for (int jj = 0; jj < 2; ++jj) {
                float vr[4], vi[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    vr[r] = fmaf(d[2 * jj][r], kd, c[2 * jj][r] * k1);
                    vi[r] = fmaf(d[2 * jj + 1][r], kd, c[2 * jj + 1][r] * k1);
                    if (!one) { vr[r] *= k2; vi[r] *= k2; }
                }
                const int oo = 128 * col + 16 * (tb + 2 * jj) + 4 * kq;
                if constexpr (SL) {
                    if (accum) { // the earlier slices' sums (the guard's output power below is the power of the SUM)
                        if (full) {
                            const auto p0 = __builtin_amdgcn_raw_buffer_load_b128(ry, oo * 8, 0, 0), p1 = __builtin_amdgcn_raw_buffer_load_b128(ry, oo * 8 + 16, 0, 0);
                            vr[0] += __uint_as_float(p0[0]); vi[0] += __uint_as_float(p0[1]); vr[1] += __uint_as_float(p0[2]); vi[1] += __uint_as_float(p0[3]);
                            vr[2] += __uint_as_float(p1[0]); vi[2] += __uint_as_float(p1[1]); vr[3] += __uint_as_float(p1[2]); vi[3] += __uint_as_float(p1[3]);
                        } else {
                            for (int r = 0; r < 4; ++r)
                                if (seg0 + oo + r < n) { const float2 pv = y[seg0 + oo + r]; vr[r] += pv.x; vi[r] += pv.y; }
                        }
                    }
                }
                if (full) { // (the guard's output power: the outputs of the span only)
#pragma unroll
                    for (int r = 0; r < 4; ++r) py = fmaf(vr[r], vr[r], fmaf(vi[r], vi[r], py));
                } else
                    for (int r = 0; r < 4; ++r)
                        if (seg0 + oo + r < n) py = fmaf(vr[r], vr[r], fmaf(vi[r], vi[r], py));
                if (full) {
                    const u32x4_h w0 = {__float_as_uint(vr[0]), __float_as_uint(vi[0]), __float_as_uint(vr[1]), __float_as_uint(vi[1])};
                    const u32x4_h w1 = {__float_as_uint(vr[2]), __float_as_uint(vi[2]), __float_as_uint(vr[3]), __float_as_uint(vi[3])};
                    __builtin_amdgcn_raw_buffer_store_b128(w0, ry, oo * 8, 0, GR4_HF_STORE_AUX);
                    __builtin_amdgcn_raw_buffer_store_b128(w1, ry, oo * 8 + 16, 0, GR4_HF_STORE_AUX);
                } else {
                    for (int r = 0; r < 4; ++r)
                        if (seg0 + oo + r < n) y[seg0 + oo + r] = make_float2(vr[r], vi[r]);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < NL4; ++u) put_next(plo, u, cur[u], s_nx);
            note(sg, slow_cur ? slow_cur : 3);
            py = __builtin_inff();
        }
        if (seg0 + 128L * col >= n) py = __builtin_inff(); // (a column past the end of the span)
        put_stats(oth, (int)(sg & 1));
        put_ypower(py, (int)(sg & 1));
        s_cur    = s_nx;
        inv_cur  = inv_nx;
        slow_cur = slow_nx;
        px_prev  = px_cur;
        px_cur   = px_nx;
        __syncthreads();
    };
    for (long sg = sfirst; sg < slast; sg += 2) {
        segment(sg, pls[0], pls[1], nxa, nxb);
        if (sg + 1 < slast) segment(sg + 1, pls[1], pls[0], nxb, nxa);
    }
    if (guard) judge(slast - 1, px_prev);
    if (new_hist != nullptr && blockIdx.x == 0) {
        for (int h = tid; h < Kh; h += 256) {
            const long i = n - Kh + h;
            new_hist[h]  = i >= 0 ? x[i] : hist[Kh + i];
        }
    }
}

// the table the kernel reads: per channel a block of hf_block_units(KS) 16-bit units (see there).  Fragment (plane p, K-step ks, lane l, element t) = tap-plane value
// b_p[Hb + (l & 15) - (32 ks + 8 (l >> 4) + t)], b_1 = f16(b t), b_2 = f16((b t - b_1) 2^11), t = the power of two that puts the channel's largest tap in [2^14, 2^15).
// false: taps this form cannot carry (non-finite) -- the caller keeps its other kernels
bool fir_f16_make_afrag(const float* taps_all, size_t ntaps, int* KS_out, std::vector<unsigned short>* af, size_t nch, int force_ks) {
    const int KS = force_ks ? force_ks : std::max(3, (int)((ntaps - 1 + 16 + 31) / 32)), Hb = 32 * KS - 16;
    if (KS < 3 || KS > 9 || (size_t)Hb + 1 < ntaps) return false;
    const size_t units = (size_t)hf_block_units(KS);
    af->assign(nch * units, 0);
    for (size_t c = 0; c < nch; ++c) {
        const float*    taps = taps_all + c * ntaps;
        unsigned short* blk  = af->data() + c * units;
        unsigned        mx   = 0;
        for (size_t k = 0; k < ntaps; ++k) {
            unsigned u;
            std::memcpy(&u, &taps[k], 4);
            mx = std::max(mx, u & 0x7fffffffu);
        }
        if (mx >= 0x7f800000u) return false;
        const int      e  = std::min(std::max((int)(mx >> 23), 15), 254);
        const unsigned tb = (unsigned)(268 - e) << 23, ib = (unsigned)(e - 14) << 23;
        float          t, inv_t;
        std::memcpy(&t, &tb, 4);
        std::memcpy(&inv_t, &ib, 4);
        std::vector<unsigned short> pl[3];
        for (auto& v : pl) v.assign(ntaps, 0);
        for (size_t k = 0; k < ntaps; ++k) {
            const float          b  = taps[k] * t;
            const unsigned short h  = host_f16_rne(b);
            const float          r1 = (b - host_f16_to_f(h)) * 2048.f; // (exact in float32)
            const unsigned short m  = host_f16_rne(r1);
            pl[0][k] = h;
            pl[1][k] = m;
            pl[2][k] = host_f16_rne((r1 - host_f16_to_f(m)) * 2048.f);
        }
        for (int p = 0; p < 3; ++p)
            for (int ks = 0; ks < KS; ++ks)
                for (int l = 0; l < 64; ++l)
                    for (int tt = 0; tt < 8; ++tt) {
                        const int k = Hb + (l & 15) - (32 * ks + 8 * (l >> 4) + tt);
                        if (k >= 0 && (size_t)k < ntaps) blk[(((size_t)p * KS + ks) * 64 + l) * 8 + tt] = pl[p][k];
                    }
        const int nt = (int)ntaps;
        double    h2 = 0;
        for (size_t k = 0; k < ntaps; ++k) h2 += (double)taps[k] * taps[k];
        const float gthr = (float)(h2 / 128.0); // (fir.hip, kGuardSegmentRatio)
        std::memcpy(blk + KS * 1536, &inv_t, 4);
        std::memcpy(blk + KS * 1536 + 2, &nt, 4);
        std::memcpy(blk + KS * 1536 + 4, &gthr, 4);
        std::memcpy(blk + KS * 1536 + 8, taps, ntaps * sizeof(float));
    }
    *KS_out = KS;
    return true;
}

// y[i] = sum_k b[k] x[i - delay - k] (+ y[i] when accum), i < n; hist = the Kh samples in front of x; x and y 16-byte aligned, strides multiples of 4
// flags: nch x ceil(n / 4096) bytes (channel c at flags + c * flags_stride): the segments fir_exact_launch evaluates again behind the (last) launch
int fir_f16_launch(int KS, const float* x, long n, const float* hist, int Kh, const void* table, float* y, hipStream_t st, float* new_hist, long in_stride, long out_stride, unsigned nch, int delay, int accum,
                   int guard, unsigned char* flags, long flags_stride, float gthr) {
    if (KS < 3 || KS > 9 || 32 * KS - 16 + delay > kHfSeg) return GR4HIP_UNSUPPORTED;
    const auto tb   = static_cast<const unsigned short*>(table);
    const long nseg = ceil_div(n, (long)kHfSeg);
    const long wgs  = KS <= 5 ? 2 * GR4_F16_TARGET_WGS : GR4_F16_TARGET_WGS; // (narrow windows: shorter runs measured +3 % -- 64 taps 620 -> 642 Gsamples/s --, wide ones -1 %)
    const int  spw  = (int)std::min<long>(std::max<long>(nseg * (long)nch / wgs, 1), GR4_F16_MAX_SPW); // segments per workgroup: the prologue (tap fragments, first staging) once per run
    const dim3 grid((unsigned)ceil_div(nseg, (long)spw), nch);
#define GR4_HF_CASE(K) case K: hipLaunchKernelGGL(fir_mfma_f16x2_kernel<K>, grid, dim3(256), 0, st, x, hist, Kh, tb, y, n, new_hist, in_stride, out_stride, delay, accum, spw, guard, flags, flags_stride, gthr); break
    switch (KS) {
        GR4_HF_CASE(3);
        GR4_HF_CASE(4);
        GR4_HF_CASE(5);
        GR4_HF_CASE(6);
        GR4_HF_CASE(7);
        GR4_HF_CASE(8);
        GR4_HF_CASE(9);
    default: return GR4HIP_UNSUPPORTED;
    }
#undef GR4_HF_CASE
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

// the same on complex samples (real taps); hist = the Kh complex samples in front of x; x and y 16-byte aligned; `table` from fir_f16_make_afrag (one channel)
// delay / accum / gthr: a slice of a longer filter (y[i] (+)= sum_k b[k] x[i - delay - k]; gthr > 0: the whole filter's threshold, judged on the sums this pass leaves)
int fir_f16_c32_launch(int KS, const float* x, long n, const float* hist, int Kh, const void* table, float* y, hipStream_t st, float* new_hist, int guard, unsigned char* flags /*ceil(n / 2048) bytes*/,
                       int delay, int accum, float gthr) {
    if (KS < 3 || KS > 9 || 32 * KS - 16 + delay > kHfSegC) return GR4HIP_UNSUPPORTED;
    const bool sl = delay != 0 || accum != 0 || gthr > 0.f;
    const auto tb   = static_cast<const unsigned short*>(table);
    const auto xc   = reinterpret_cast<const float2*>(x), hc = reinterpret_cast<const float2*>(hist);
    const auto yc   = reinterpret_cast<float2*>(y), nh = reinterpret_cast<float2*>(new_hist);
    const long nseg = ceil_div(n, (long)kHfSegC);
    const int  spw  = (int)std::min<long>(std::max<long>(nseg / GR4_F16_TARGET_WGS, 1), GR4_F16_MAX_SPW);
    const dim3 grid((unsigned)ceil_div(nseg, (long)spw));
#define GR4_HFC_CASE(K) case K: if (sl) hipLaunchKernelGGL((fir_mfma_f16x2_c32_kernel<K, true>), grid, dim3(256), 0, st, xc, hc, Kh, tb, yc, n, nh, spw, guard, flags, delay, accum, gthr); \
                                else hipLaunchKernelGGL((fir_mfma_f16x2_c32_kernel<K, false>), grid, dim3(256), 0, st, xc, hc, Kh, tb, yc, n, nh, spw, guard, flags, 0, 0, 0.f); break
    switch (KS) {
        GR4_HFC_CASE(3);
        GR4_HFC_CASE(4);
        GR4_HFC_CASE(5);
        GR4_HFC_CASE(6);
        GR4_HFC_CASE(7);
    case 8: // (no sliced instantiation: it would keep two registers in scratch -- fir.hip gives such a slice the 9-step kernel on zero-padded taps)
        if (sl) return GR4HIP_UNSUPPORTED;
        hipLaunchKernelGGL((fir_mfma_f16x2_c32_kernel<8, false>), grid, dim3(256), 0, st, xc, hc, Kh, tb, yc, n, nh, spw, guard, flags, 0, 0, 0.f);
        break;
        GR4_HFC_CASE(9);
    default: return GR4HIP_UNSUPPORTED;
    }
#undef GR4_HFC_CASE
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

} // namespace gr4
