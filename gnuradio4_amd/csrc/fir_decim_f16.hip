// fir_decim_f16.hip -- BasicDecimatingFilter<float>, decimate by D = 8 (up to 1025 taps: BASELINE configs[2]), 16 (897) or 32 (641), on the f16 matrix pipe: the band form of the
// contraction (written out below for D = 8; a segment is 8192 input samples whatever D: 16 columns of 512 samples, 32 / D tile rows per column, D / 2 K-steps apart)
//
//     y[m] = sum_k b[k] x[D m - k]:   D[j][c] = sum_u A[j][u] B[u][c],   A[j][u] = b[Hb + D j - u],   B[u][c] = staged[base(tile) + u],   u < 128 KQ
//
// (samples in stream order, the decimation in the tap operand: a tile of 16 outputs sees a window of Hb + 121 <= 128 KQ samples, the next tile starts 128 samples on) with
// fir_f16.hip's arithmetic -- samples and taps as two f16 terms under a per-segment block exponent, three products per tap -- and its safeguards: every segment's statistics
// (largest magnitude, quietest group of four, power) decide its block scale and whether it is given to the f16 pipe at all, every segment's output power is judged against
// its input power, and a rejected segment is evaluated again with three-term f16 products (float32 products) at the end of the workgroup's run.
//
// What the bf16 split-K kernel of fir_bf16.hip paid for (305 G input samples/s at 1024 taps): six products, three planes, two tiles per segment.  Here the K-steps are split
// over the four waves of a workgroup as there -- a wave keeps the tap fragments of its quarter (KQ K-steps: 72 registers at KQ = 9) -- but a segment is 1024 outputs
// = 64 tiles (16 columns of 64 outputs, four tile rows), a wave runs its K quarter for ALL four tile rows, and because tile rows sit exactly four K-steps apart it walks ONE
// sample-fragment stream of KQ + 12 fragments for the 4 KQ K-steps it evaluates (a third of the LDS operand reads).  The four partial tiles meet in LDS; thread (w', lane)
// sums tile row w' and takes it out.  108 f16 MFMAs per wave and 8192 input samples: half the matrix-pipe work per input sample of the 256-tap FIR of fir_f16.hip.
#ifndef GR4_DH_LOAD_AUX // cache policy of the sample loads (2 = nt) / result stores: developer builds override; profiles/r05_streaming_hints.txt
#define GR4_DH_LOAD_AUX 0
#endif
#ifndef GR4_DH_STORE_NT
#define GR4_DH_STORE_NT 0
#endif
#include "common.hpp"
#include "buffer_ops.hpp"
#include "fir_f16_common.hpp"
#include "fir_band_hooks.hpp"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <type_traits>

namespace gr4 {

constexpr int kDhSegIn = 8192; // a segment: 8192 input samples = 8192 / D outputs = 16 columns of 512 samples, 32 / D tile rows per column (D = 4, 8, 16, 32)

// the table fir_decim_f16_make_table writes, in 16-bit units: [4 waves][3 planes][KQ][64 lanes][8] f16 fragments, 8 units of header {float 1 / t, int ntaps, float guard
// threshold, -}, 1040 float taps
__host__ __device__ constexpr int dh_frag_units(int KQ) { return 4 * 3 * KQ * 512; }
__host__ __device__ constexpr int dh_table_units(int KQ) { return dh_frag_units(KQ) + 8 + 2 * 1040; }

// the matrix-pipe evaluation of a staged segment (two terms per factor, three products): this wave's K quarter for the tile rows, its partial tiles to `part`
template <int D, int KQ, int PL>
__device__ __forceinline__ void dh_contract(const u32x4_h (&a)[2][KQ], const unsigned short* __restrict__ pls, float (*__restrict__ part)[32 / D][64][4], int wave, int lane) {
    constexpr int TR = 32 / D, TS = D / 2, NM = KQ + TS * (TR - 1); // tile rows per column, K-steps between them (16 D samples), fragments of the stream
    const int col = lane & 15, kq = lane >> 4;
    auto      P   = [](int s_) { return s_ + 8 * (s_ >> 9); };
    f32x4_h   c[TR], d[TR];
#pragma unroll
    for (int j = 0; j < TR; ++j) c[j] = d[j] = f32x4_h{0.f, 0.f, 0.f, 0.f};
#ifdef GR4_T_DH_MFMA32 // timing-only build (results are wrong; profiles/r06_decim_floor.txt): the products of two tile rows as ONE 32 x 32 x 16 instruction on the same operand
                       // registers -- half the matrix instructions and operand reads for the same multiply-adds = the bound of a kernel re-tiled for 32 x 32 x 16
    using f32x16_h = __attribute__((ext_vector_type(16))) float;
    f32x16_h c32[(TR + 1) / 2], d32[(TR + 1) / 2];
#pragma unroll
    for (int j = 0; j < (TR + 1) / 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) c32[j][e] = d32[j][e] = 0.f;
#endif
    const int sb = 512 * col + 32 * KQ * wave + 8 * kq;
#pragma unroll
    for (int m = 0; m < NM; ++m) {
        const int     qo = P(sb + 32 * m);
        const f16x8_h b1 = *reinterpret_cast<const f16x8_h*>(pls + qo), b2 = *reinterpret_cast<const f16x8_h*>(pls + PL + qo);
#pragma unroll
        for (int tr = 0; tr < TR; ++tr) {
            const int ks = m - TS * tr;
            if (ks < 0 || ks >= KQ) continue;
            const f16x8_h a1 = __builtin_bit_cast(f16x8_h, a[0][ks]), a2 = __builtin_bit_cast(f16x8_h, a[1][ks]);
#ifdef GR4_T_DH_NOMFMA // timing-only builds (tools/ab_dh.sh, profiles/r05_decim_bounds.txt): the operand reads without the products
            asm volatile("" ::"v"(a1), "v"(a2), "v"(b1), "v"(b2));
#elif defined(GR4_T_DH_MFMA32)
            if ((tr & 1) == 0) {
                c32[tr >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b1, c32[tr >> 1], 0, 0, 0);
                d32[tr >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a1, b2, d32[tr >> 1], 0, 0, 0);
                d32[tr >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a2, b1, d32[tr >> 1], 0, 0, 0);
            } else {
                asm volatile("" ::"v"(a1), "v"(a2)); // (the odd tile rows' tap fragments stay live: a re-tiled kernel holds as many)
            }
#else
            c[tr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1, c[tr], 0, 0, 0);
            d[tr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2, d[tr], 0, 0, 0);
            d[tr] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1, d[tr], 0, 0, 0);
#endif
        }
    }
#ifdef GR4_T_DH_MFMA32
#pragma unroll
    for (int tr = 0; tr < TR; ++tr)
#pragma unroll
        for (int e = 0; e < 4; ++e) { c[tr][e] = c32[tr >> 1][8 * (tr & 1) + e] + c32[tr >> 1][8 * (tr & 1) + 4 + e]; d[tr][e] = d32[tr >> 1][8 * (tr & 1) + e] + d32[tr >> 1][8 * (tr & 1) + 4 + e]; }
#endif
#pragma unroll
    for (int tr = 0; tr < TR; ++tr)
        *reinterpret_cast<float4*>(&part[wave][tr][lane][0]) =
            make_float4(c[tr][0] + d[tr][0] * (1.f / 2048.f), c[tr][1] + d[tr][1] * (1.f / 2048.f), c[tr][2] + d[tr][2] * (1.f / 2048.f), c[tr][3] + d[tr][3] * (1.f / 2048.f));
}

// HOOK (round 5): the filter's per-sample neighbours ride in this launch like in the bf16 band kernels (fir_band_hooks.hpp): `hk.pre` on every sample of x on its way into
// the statistics and the f16 planes (the carried history holds prologue OUTPUTS and is taken as it lies; the next history this launch writes is made of them too), `hk.post`
// on every output before its store (the verdict's output power is the FILTER's, in front of it).  The channeliser -- rotator -> decimate-by-8 complex FIR -- is this shape.
// HOOK == 2 (round 5): the load program is ONE rotator on a complex stream and there is no store program -- rotator -> decimating FIR, a down-converter -- : no program walk,
// the phase of a lane's first sample from one 64-bit product per segment, every further one by an integer addition (ewise.hpp: bit-identical to HOOK == 1).
template <int D, int KQ, int HOOK> // decimation 4 / 8 / 16 / 32; K-steps of 32 per wave: window 128 KQ samples, Hb = 128 KQ - 16 D samples in front of a tile's first output
__global__ __launch_bounds__(256, 2) void fir_decim_f16x2_kernel(const float* __restrict__ x, const float* __restrict__ hist /*hist[h] = x[-Kh + h]*/, int Kh,
                                                                  const unsigned short* __restrict__ tab, float* __restrict__ y, long n_out, long n_in,
                                                                  float* __restrict__ new_hist, int guard, int seg_per_wg /*<= kDhMaxSpw*/,
                                                                 unsigned char* __restrict__ flags /*one byte per segment: what fir_exact_kernel evaluates again behind this launch*/,
                                                                 int cplx /*the streams are complex<float> read as floats (n_in, n_out, Kh in floats; D complex samples in per complex sample out): the tap table carries the interleaving*/,
                                                                 BdHooks hk) {
    constexpr int TR = 32 / D, SO = kDhSegIn / D, Hb = 128 * KQ - 16 * D, NS = kDhSegIn + Hb; // tile rows per column, outputs per segment, staged samples per segment (a multiple of 128)
    static_assert(Hb > 0, "the window must hold a tile's 16 D input samples");
    constexpr int PL  = NS + 8 * (NS / 512 + 1) + 16;      // f16 elements per plane: one 16-byte chunk of padding per 512 samples (the 16 columns of a fragment read are 512 samples apart)
    constexpr int NL4 = (NS / 4 + 255) / 256;              // float4 loads a lane holds for the next segment
    const u32x4_h* afrag = reinterpret_cast<const u32x4_h*>(tab);
    const float    inv_t = *reinterpret_cast<const float*>(tab + dh_frag_units(KQ));
    const float    gthr  = *reinterpret_cast<const float*>(tab + dh_frag_units(KQ) + 4);
    extern __shared__ __attribute__((aligned(16))) unsigned short pls[]; // [2][PL]: planes x1, x2
    __shared__ __attribute__((aligned(16))) float          part[4][TR][64][4]; // [K quarter = wave][tile row][lane][row within the lane's four]
    __shared__ __attribute__((aligned(16))) unsigned       stat[12];
    __shared__ __attribute__((aligned(16))) float          ystat[4][16]; // per wave: the output powers of the tile rows it takes out, per column
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
    auto P = [](int s_) { return s_ + 8 * (s_ >> 9); };

    auto xs = [&](long i) __attribute__((always_inline)) -> float { return i >= 0 ? (i < n_in ? x[i] : 0.f) : (i >= -(long)Kh ? hist[Kh + i] : 0.f); };
    float4 nxt[NL4];
    auto   load_next = [&](long sg) __attribute__((always_inline)) { // sg >= 1: nothing below 0
        const long   i0   = sg * kDhSegIn - Hb;
        const long   nrec = n_in - i0 < (long)NS ? n_in - i0 : (long)NS;
        const rsrc_t r    = make_rsrc(x + i0, (unsigned)(nrec > 0 ? nrec * 4 : 0));
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, 256 * u * 16, GR4_DH_LOAD_AUX);
            nxt[u]       = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto load_general = [&](long sg) __attribute__((always_inline)) {
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const int q = tid + 256 * u;
            const long i0 = sg * kDhSegIn - Hb + 4L * q;
            nxt[u] = q < NS / 4 ? make_float4(xs(i0), xs(i0 + 1), xs(i0 + 2), xs(i0 + 3)) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto load_seg = [&](long sg) __attribute__((always_inline)) {
        if (sg > 0) load_next(sg);
        else load_general(0);
    };
    auto hook_loaded = [&](long sg) __attribute__((always_inline)) { // the prologue on the samples of x among the loaded ones (sg = 0: the history in front of them stays as it lies)
        if constexpr (HOOK == 2) {
            // nxt[u] = complex samples kc, kc + 1 with kc = (sg kDhSegIn - Hb) / 2 + 2 (tid + 256 u) of this span: rotor index pos + kc + 1.  Samples past the span's end are
            // zeros and stay zeros under a finite rotor; the history in front of position 0 (sg == 0 only) stays as it lies
            BdRotor rot = bd_rotor_start(hk.pre, sg * kDhSegIn - Hb, tid);
#pragma unroll
            for (int u = 0; u < NL4; ++u) {
                const float4 v = nxt[u];
                float4       w = bd_rotor_next(v, rot);
                if (sg == 0) {
                    const long fi = 4L * (tid + 256 * u) - Hb;
                    if (fi < 0) { w.x = v.x; w.y = v.y; }
                    if (fi + 2 < 0) { w.z = v.z; w.w = v.w; }
                }
                nxt[u] = w;
            }
        } else if constexpr (HOOK == 1) {
            if (hk.pre.n_ops > 0) {
#pragma unroll
                for (int u = 0; u < NL4; ++u) {
                    const int  q  = tid + 256 * u;
                    const long fi = sg * kDhSegIn - Hb + 4L * q;
                    if (q < NS / 4 && fi + 3 >= 0 && fi < n_in) {
                        const float4 w = bd_hook4(nxt[u], hk.pre, cplx, fi); // (fi, Kh and n_in are even for complex streams: a pair never straddles position 0)
                        float4       v = nxt[u];
                        if (fi >= 0) v = w;
                        else if (fi + 2 >= 0) { v.z = w.z; v.w = w.w; if (!cplx && fi + 1 >= 0) v.y = w.y; }
                        else if (!cplx) v.w = w.w;
                        if (fi + 3 >= n_in) { // (the span's end inside this chunk: zeros stay zeros)
                            if (fi + 1 >= n_in) v.y = 0.f;
                            if (fi + 2 >= n_in) v.z = 0.f;
                            v.w = 0.f;
                        }
                        nxt[u] = v;
                    }
                }
            }
        }
    };
    auto put_stats = [&]() __attribute__((always_inline)) {
        float    mf = 0.f, px = 0.f;
        unsigned mn = 0xffffffffu;
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const float m4 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(nxt[u].x), __builtin_fabsf(nxt[u].y)), __builtin_fmaxf(__builtin_fabsf(nxt[u].z), __builtin_fabsf(nxt[u].w)));
            mf = __builtin_fmaxf(mf, m4);
            mn = min(mn, __float_as_uint(m4) - 1u);
            px = fmaf(nxt[u].x, nxt[u].x, fmaf(nxt[u].y, nxt[u].y, fmaf(nxt[u].z, nxt[u].z, fmaf(nxt[u].w, nxt[u].w, px))));
        }
        unsigned mx = __float_as_uint(mf);
        mx = hf_wave_reduce_u32(mx, [](unsigned a_, unsigned b_) { return a_ > b_ ? a_ : b_; });
        mn = hf_wave_reduce_u32(mn, [](unsigned a_, unsigned b_) { return a_ < b_ ? a_ : b_; });
        px = hf_wave_sum(px);
        if (lane == 0) { stat[wave] = mx; stat[4 + wave] = mn; stat[8 + wave] = __float_as_uint(px); }
    };
    auto block_scale = [&](float& s, float& inv_s, float& px) __attribute__((always_inline)) -> int { // 0: the f16 pipe; 1: the spread; 2: a non-finite sample
        const uint4 m4 = *reinterpret_cast<const uint4*>(&stat[0]), n4 = *reinterpret_cast<const uint4*>(&stat[4]), p4 = *reinterpret_cast<const uint4*>(&stat[8]);
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
    // thread (w', lane) takes tile row w' out: the four K quarters' partial tiles summed in a fixed order, the block scales off, y[seg + 16 TR col + 16 w' + 4 kq + r]
    auto take_out = [&](long sg, float k, float& py) __attribute__((always_inline)) {
        // thread (w', lane) takes tile rows w', w' + 4, ... out (D = 4: two each; D = 16 / 32 have two / one tile row: the other waves take nothing)
#pragma unroll
        for (int tr = wave; tr < TR; tr += 4) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = ((part[0][tr][lane][r] + part[1][tr][lane][r]) + (part[2][tr][lane][r] + part[3][tr][lane][r])) * k;
            const long o = sg * SO + (long)(16 * TR) * col + 16 * tr + 4 * kq;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (o + r < n_out) py = fmaf(v[r], v[r], py); // (the filter's output, in front of the store program)
            if constexpr (HOOK == 1) {
                if (hk.post.n_ops > 0) { const float4 w = bd_hook4(make_float4(v[0], v[1], v[2], v[3]), hk.post, cplx, o); v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w; }
            }
            if (o + 3 < n_out) {
#if GR4_DH_STORE_NT
                __builtin_nontemporal_store(f32x4_h{v[0], v[1], v[2], v[3]}, reinterpret_cast<f32x4_h*>(y + o));
#else
                *reinterpret_cast<float4*>(y + o) = make_float4(v[0], v[1], v[2], v[3]);
#endif
            }
            else {
                for (int r = 0; r < 4; ++r)
                    if (o + r < n_out) y[o + r] = v[r];
            }
        }
    };
    u32x4_h a[2][KQ]; // this wave's quarter of the tap fragments (planes 0, 1 of the table's three)
#pragma unroll
    for (int p = 0; p < 2; ++p)
#pragma unroll
        for (int ks = 0; ks < KQ; ++ks) a[p][ks] = afrag[((wave * 3 + p) * KQ + ks) * 64 + lane];
    // the guard (fir_f16.hip): sixteen times the power of the QUIETEST of the segment's sixteen output columns, at the input rate, against 2^-12 (sum b^2) x the input power
    auto rejected = [&](float px) __attribute__((always_inline)) -> bool {
        const float pc = (ystat[0][col] + ystat[1][col]) + (ystat[2][col] + ystat[3][col]); // (a wave's entry is the sum over the tile rows it took out)
        return __builtin_amdgcn_readfirstlane((int)(px < 0.f || 16.f * hf_row_min(pc) * (float)D < gthr * px)) != 0; // (px < 0: block_scale could not form the power)
    };
    const long nseg = (n_out + SO - 1) / SO, sfirst = (long)blockIdx.x * seg_per_wg, slast = sfirst + seg_per_wg < nseg ? sfirst + seg_per_wg : nseg;
    if (sfirst >= slast) return;
    if (tid < slast - sfirst) flags[sfirst + tid] = 0; // segments this kernel cannot vouch for are MARKED (1: spread beyond the block exponent; 2: a non-finite sample; 3: rejected
                                                       // by the guard); fir_exact_kernel (fir_exact.hip), launched behind this one, evaluates them again on the FP64 matrix pipe
    {
        load_seg(sfirst);
        float px_prev = 0.f;
        int   kind_prev = -1; // (nothing to judge yet)
        for (long sg = sfirst; sg < slast; ++sg) {
            hook_loaded(sg);
#ifndef GR4_T_DH_NOSTATS // timing-only: whatever the statistics words hold
            put_stats();
#endif
            __syncthreads(); // the statistics are complete; every wave is done with the planes, the partial tiles and the verdict words of the segment before
            float     s, inv_s, px;
            const int kind = block_scale(s, inv_s, px);
            if (guard && kind_prev == 0 && rejected(px_prev) && tid == 0) flags[sg - 1] = 3; // the verdict on the segment before (its output powers: a barrier ago)
#pragma unroll
            for (int u = 0; u < NL4; ++u) {
                const int q = tid + 256 * u;
                if (256 * (u + 1) <= NS / 4 || q < NS / 4) {
                    unsigned h0, l0, h1, l1;
#ifdef GR4_T_DH_NOSPLIT // timing-only: the samples' bits
                    h0 = __float_as_uint(nxt[u].x), l0 = __float_as_uint(nxt[u].y), h1 = __float_as_uint(nxt[u].z), l1 = __float_as_uint(nxt[u].w);
#else
                    hf_split2(nxt[u].x, nxt[u].y, s, h0, l0);
                    hf_split2(nxt[u].z, nxt[u].w, s, h1, l1);
#endif
                    *reinterpret_cast<uint2*>(pls + P(4 * q))      = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(pls + PL + P(4 * q)) = make_uint2(l0, l1);
                }
            }
#ifndef GR4_T_DH_NOLOAD // timing-only: every segment is the run's first again
            if (sg + 1 < slast) load_next(sg + 1);
#endif
            __syncthreads();
#ifdef GR4_T_DH_NOCONTRACT // timing-only: no operand reads, no products, no partial tiles
            if (kind == 0 && n_out < 0) dh_contract<D, KQ, PL>(a, pls, part, wave, lane);
#else
            if (kind == 0) dh_contract<D, KQ, PL>(a, pls, part, wave, lane);
#endif
            else if (tid == 0) flags[sg] = (unsigned char)kind;
            __syncthreads();
            float py = 0.f;
#ifdef GR4_T_DH_NOOUT // timing-only: nothing leaves
            if (kind == 0 && n_out < 0) take_out(sg, inv_t * inv_s, py);
#else
            if (kind == 0) take_out(sg, inv_t * inv_s, py);
#endif
            if (wave >= TR) py = 0.f;                                                                  // (no tile row of its own: nothing to add to the columns' sums)
            else if (kind != 0 || sg * SO + (long)(16 * TR) * col >= n_out) py = __builtin_inff(); // (nothing to judge; a column past the end of the span)
            py = hf_column_sum(py);
            if (lane < 16) ystat[wave][lane] = py;
            px_prev   = px;
            kind_prev = kind;
        }
        __syncthreads();
        if (guard && kind_prev == 0 && rejected(px_prev) && tid == 0) flags[slast - 1] = 3;
    }
    if (new_hist != nullptr && blockIdx.x == 0) bd_new_hist<HOOK != 0>(x, hist, Kh, n_in, new_hist, tid, hk);
}

// the table of fir_decim8_f16x2_kernel<KQ> (see dh_table_units): fragment (wave w, plane p, K-step ks, lane l, element t) = tap-plane value
// b_p[Hb + 8 (l & 15) - (32 (KQ w + ks) + 8 (l >> 4) + t)]; planes as fir_f16_make_afrag's.  false: taps or a shape this kernel does not carry
// cplx: complex<float> streams read as floats (fir_bf16.hip's scheme): row j of a tile -- 16 float outputs = 8 complex ones, 16 D floats of input further on per tile, exactly
// the float geometry -- sits s_j = 2 D (j >> 1) + (j & 1) floats into the window and sees tap k at window position Hb + s_j - 2 k: A[j][u] = b[(Hb + s_j - u) / 2] where that is
// even, else 0 (half of the matrix pipe's work multiplies zeros; the launches are bound by the stream)
bool fir_decim_f16_make_table(const float* taps, size_t ntaps, size_t D, int* KQ_out, std::vector<unsigned short>* tab, bool cplx) {
    if ((D != 4 && D != 8 && D != 16 && D != 32) || ntaps < 2) return false;
    int KQ = 0;
    for (int k : {3, 5, 7, 9})
        if (128 * k > 16 * (int)D && (size_t)(128 * k - 16 * (int)D + 1) >= (cplx ? 2 * ntaps - 1 : ntaps)) { KQ = k; break; } // Hb = 128 KQ - 16 D >= taps - 1 (complex: 2 (taps - 1))
    if (!KQ) return false;
    const int Hb = 128 * KQ - 16 * (int)D;
    unsigned  mx = 0;
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
    double h2 = 0;
    for (size_t k = 0; k < ntaps; ++k) {
        const float          b  = taps[k] * t;
        const unsigned short h  = host_f16_rne(b);
        const float          r1 = (b - host_f16_to_f(h)) * 2048.f;
        const unsigned short m  = host_f16_rne(r1);
        pl[0][k] = h;
        pl[1][k] = m;
        pl[2][k] = host_f16_rne((r1 - host_f16_to_f(m)) * 2048.f);
        h2 += (double)taps[k] * taps[k];
    }
    tab->assign((size_t)dh_table_units(KQ), 0);
    for (int w = 0; w < 4; ++w)
        for (int p = 0; p < 3; ++p)
            for (int ks = 0; ks < KQ; ++ks)
                for (int l = 0; l < 64; ++l)
                    for (int tt = 0; tt < 8; ++tt) {
                        const int j = l & 15;
                        long      k = (long)Hb + (cplx ? 2L * (long)D * (j >> 1) + (j & 1) : (long)D * j) - (32 * (KQ * w + ks) + 8 * (l >> 4) + tt);
                        if (cplx) k = (k & 1) ? -1 : k / 2;
                        if (k >= 0 && (size_t)k < ntaps) (*tab)[((((size_t)w * 3 + p) * KQ + ks) * 64 + l) * 8 + tt] = pl[p][(size_t)k];
                    }
    unsigned short* hd   = tab->data() + dh_frag_units(KQ);
    const int       nt   = (int)ntaps;
    // P_y D < 2^-6 (sum b^2) P_x: 18 dB more rejected than white noise would lose.  (2^-7 -- fir.hip's kGuardSegmentRatio -- until round 6's last day: the 22-bit products' error is
    // coherent on a tone, and a tone in the transition band of a long decimating filter, 18 .. 21 dB above what passes, left 7.9 / 9.4e-6 of the output's rms at decimation 8 / 16 in
    // 1 300 targeted streams, tools/dbg/fir_coherent.py: inside the bar, too close to it.  3 dB earlier: a factor 1.4 on that error.)
    const float     gthr = (float)(h2 / 64.0);
    std::memcpy(hd, &inv_t, 4);
    std::memcpy(hd + 2, &nt, 4);
    std::memcpy(hd + 4, &gthr, 4);
    std::memcpy(hd + 8, taps, ntaps * sizeof(float));
    *KQ_out = KQ;
    return true;
}

// y[m] = sum_k b[k] x[D m - k], m < n_out = n_in / D, D = 4 / 8 / 16 / 32; hist[h] = x[-Kh + h]; x and y 16-byte aligned
template <int D>
static int fir_decim_f16_launch_d(int KQ, const float* x, long n_in, const float* hist, int Kh, const unsigned short* tb, float* y, long n_out, hipStream_t st, float* new_hist, int guard, unsigned char* flags, int cplx,
                                  const BdHooks& hk) {
    const bool hooked = hk.pre.n_ops > 0 || hk.post.n_ops > 0;
    const bool ddc    = cplx && hk.pre.rotor_only && hk.pre.n_ops == 1 && hk.post.n_ops == 0 && (Kh % 4) == 0; // rotator -> decimator: the phase stepped in integers
    if (hooked && !ddc && ((Kh % 4) != 0 || (D == 4 && KQ > 5))) return GR4HIP_UNSUPPORTED; // (fir.hip does not send these shapes here)
    static const int kSpwEnv = [] { const char* e = std::getenv("GR4HIP_DH_SPW"); return e ? std::atoi(e) : 0; }(); // developer knob
    const long nseg = ceil_div(n_out, (long)(kDhSegIn / D));
    const int  spw  = kSpwEnv ? kSpwEnv : (int)std::min<long>(std::max<long>(nseg / 512, 4), 32); // segments per workgroup: the tap fragments and the first staging once per run (2^27 inputs, D = 8: 4 / 8 / 16 / 32 / 64 segments measured 739 / 758 / 777 / 788 / 520 G at 1024 taps)
    const dim3 grid((unsigned)ceil_div(nseg, (long)spw));
#define GR4_DH_CASE(K)                                                                                                                                                   \
    case K: {                                                                                                                                                            \
        if constexpr (128 * K > 16 * D) {                                                                                                                                \
            constexpr int    NS  = kDhSegIn + 128 * K - 16 * D;                                                                                                          \
            constexpr size_t lds = (size_t)2 * (NS + 8 * (NS / 512 + 1) + 16) * sizeof(unsigned short);                                                                  \
            constexpr bool kHookFits = !(D == 4 && K > 5); /* (the hooked D = 4 kernels with 7 / 9 K-steps per wave would spill: never instantiated) */                           \
            auto kern = ddc ? fir_decim_f16x2_kernel<D, K, 2> : ((hooked && kHookFits) ? fir_decim_f16x2_kernel<D, K, kHookFits ? 1 : 0> : fir_decim_f16x2_kernel<D, K, 0>);              \
            if (lds > 48 * 1024) GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); /* (per call: the attribute is per device) */ \
            hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, hist, Kh, tb, y, n_out, n_in, new_hist, guard, spw, flags, cplx, hk);                                   \
        } else return GR4HIP_UNSUPPORTED;                                                                                                                                \
    } break
    switch (KQ) {
        GR4_DH_CASE(3);
        GR4_DH_CASE(5);
        GR4_DH_CASE(7);
        GR4_DH_CASE(9);
    default: return GR4HIP_UNSUPPORTED;
    }
#undef GR4_DH_CASE
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}
// cplx: x, y, hist are complex<float> streams passed as floats: n_in, n_out, Kh count FLOATS, D is the decimation of the complex stream
// flags: ceil(n_out / (8192 / D)) bytes: the segments (8192 / D outputs, counted as n_out is) fir_exact_launch evaluates again behind this launch
// pre / post (optional): the filter's load / store programs (positions: sample indices of x[0] / y[0])
int fir_decim_f16_launch(int D, int KQ, const float* x, long n_in, const float* hist, int Kh, const void* table, float* y, long n_out, hipStream_t st, float* new_hist, int guard, unsigned char* flags, int cplx,
                         const EwiseHook* pre, const EwiseHook* post) {
    const auto tb = static_cast<const unsigned short*>(table);
    BdHooks    hk;
    if (pre) hk.pre = *pre;
    if (post) hk.post = *post;
    hk.cplx = cplx;
    switch (D) {
    case 4: return fir_decim_f16_launch_d<4>(KQ, x, n_in, hist, Kh, tb, y, n_out, st, new_hist, guard, flags, cplx, hk);
    case 8: return fir_decim_f16_launch_d<8>(KQ, x, n_in, hist, Kh, tb, y, n_out, st, new_hist, guard, flags, cplx, hk);
    case 16: return fir_decim_f16_launch_d<16>(KQ, x, n_in, hist, Kh, tb, y, n_out, st, new_hist, guard, flags, cplx, hk);
    case 32: return fir_decim_f16_launch_d<32>(KQ, x, n_in, hist, Kh, tb, y, n_out, st, new_hist, guard, flags, cplx, hk);
    default: return GR4HIP_UNSUPPORTED;
    }
}

} // namespace gr4
