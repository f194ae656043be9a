// chain16.hip -- second-generation fused complex<float> FIR -> 8192-pt FFT -> |X|^2 kernel for gfx950: 16 waves per CU, wave-private sub-transforms.
//
// Same mathematics as chain_fused.hip (FFT(y_f) = H X + E, E = FFT of the 255-sample linear-minus-circular correction e), different
// machine mapping.  chain_fd_kernel keeps 16 points per lane (8 waves, 2 per SIMD, 255 VGPRs) and synchronises the whole workgroup six
// times per frame; what bounds it is VALU issue with only two waves per SIMD to cover every barrier and LDS round trip.
// Here a lane owns 8 points, the workgroup has 16 waves (4 per SIMD, <= 128 VGPRs), and the 8192-point transform is split
// decimation-in-frequency as 16 x 512:
//
//   X[16 k' + q] = sum_m W_512^{m k'} ( W_8192^{m q} sum_h x[m + 512 h] W_16^{h q} )            m = 0..511, h, q = 0..15
//
//   phase 0  "cross pass": thread (m, sigma) reads the 16 samples x[m + 512 h] of the frame (natural order in the LDS landing buffer, so
//            lanes read consecutive addresses), forms the radix-16 outputs q = 2 q' + sigma in registers (sum / difference of the two
//            halves, constant twiddles, one radix-8 butterfly), multiplies by W_8192^{m q} and hands value (q, m) to wave q's PRIVATE
//            4.5 KB region of the work buffer.  ONE workgroup barrier.
//   phase 1  wave q transforms its 512 points on its own: three radix-8 stages (8 x 8 x 8) with two exchanges through its private
//            region -- no barrier, not even a wait beyond the LDS counter: the LDS executes one wave's accesses in order.  Times H.
//   phase 2  the correction: e needs no cross pass at all (only h = 0 is non-zero, so every q sees e[m] itself), wave q multiplies by
//            W_8192^{m q} and runs the same private transform with a half-pruned first stage (m < 256).  |H X + E|^2.
//   output   bins 16 k' + q of one wave are 64 bytes apart in HBM, so |Y|^2 goes back into the wave's private region and is picked up
//            after the next frame's top barrier by the threads that store 256 contiguous bytes per wave instruction.
//
// e = T(b) d on the f32 MFMA units exactly as in chain_fd_kernel (block-Toeplitz, 128 v_mfma_f32_16x16x4_f32 per frame), issued by waves
// 0..7 right after the barrier while waves 8..15 start their transforms; waves 8..15 fold the four K-partial tiles afterwards.
// Two hard barriers per frame (frame landed; cross pass done) instead of six; the two remaining cross-wave dependencies (private region
// free for the next cross pass; e complete) are split barriers on LDS counters: arrive early, wait late, nobody actually waits.
//
// One landing buffer is enough: it is dead after the cross pass, i.e. the LDS-DMA of the next frame has ~85 % of a frame time to land.
#include "common.hpp"
#include "buffer_ops.hpp"
#include "fft_radix.hpp"
#include "wave16_common.hpp"

#include <cmath>
#include <cstdlib>

namespace gr4 {

constexpr int kN16   = 8192;
constexpr int kT16   = 1024;                    // lanes per workgroup (16 waves, 4 per SIMD)
constexpr int kRS    = 4616;                    // bytes per wave-private region: 576 float2 + 8 (kRS / 4 = 2 mod 32: the output pick-up is conflict-free)
constexpr int kOffW  = 65536;                   // landing buffer: float2[8192] at 0
constexpr int kOffP  = kOffW + 16 * kRS;        // partial e tiles: float [4 K quarters][re, im][256]
constexpr int kOffD  = kOffP + 8192;            // Dre | Dim: planar, one pad float per 16 (conflict-free MFMA B operand), 2 x 544 floats
constexpr int kDPad16 = 544;
constexpr int kOffH  = kOffD + 2 * kDPad16 * 4; // taps: 272 floats (zero from ntaps on)
constexpr int kOffE  = kOffH + 272 * 4;         // e: float2[256]
constexpr int kOffT  = kOffE + 2048;            // the 256 samples before the frame: float2[256]
constexpr int kOffC  = kOffT + 2048;            // split-barrier counters
constexpr int kLds16 = kOffC + 16;
static_assert(kLds16 <= 160 * 1024, "LDS budget of one CU");
static_assert(kOffT % 16 == 0 && kOffW % 16 == 0, "LDS-DMA destinations are 16-byte aligned");

struct Chain16Args {
    const float2* x;      // frames * 8192 samples
    const float2* hist;   // 256 samples preceding x
    const float2* twX;    // [1024][8]  W_8192^{m (2 q' + sigma)}, thread = 512 sigma + m
    const float2* tw1;    // [64][8]    W_512^{l k}
    const float2* tw2;    // [8][8]     W_64^{l0 k}
    const float2* twE;    // [16][64][4] W_8192^{(64 r + l) q}
    const float2* Hq;     // [16][64][8] H[16 (ka + 8 kb0 + 64 kb1) + q], lane = 8 ka + kb0
    const float*  taps;   // 256 (zero padded)
    float*        out;    // frames * 8192 |Y|^2
    long          n_frames;
    unsigned long long* dbg; // GR4_C16_TIMING only
};

// the 64 KB frame: 64 one-KiB pieces, 4 per wave (pieces I0..I1-1 of this wave); natural order
template <int I0, int I1>
__device__ __forceinline__ void dma16_frame(const float2* __restrict__ xf, unsigned lds_base, int wave, int lane) {
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        const int p = 4 * wave + i;
        dma16_1k(xf + 128 * p + 2 * lane, lds_base + 1024u * (unsigned)p);
    }
}

// developer instrumentation (-DGR4_C16_TIMING): every wave stamps s_memtime at phase boundaries into a.dbg[iteration][workgroup][wave][16]
#ifdef GR4_C16_TIMING
#define G16_STAMP(i)                                                                                  \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
        if ((threadIdx.x & 63) == 0 && a.dbg) a.dbg[(((unsigned long long)it * gridDim.x + blockIdx.x) * 16 + (threadIdx.x >> 6)) * 16 + (i)] = t_; \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
#else
#define G16_STAMP(i) do { } while (0)
#endif

enum { k16Mag2 = 0, k16FftMag2 = 1 };

template <int MODE>
__global__ __launch_bounds__(kT16) void chain16_kernel(Chain16Args a) {
    constexpr bool FFTONLY = MODE == k16FftMag2;
    extern __shared__ __attribute__((aligned(16))) char smem16[]; // the ONLY LDS object
    float2*   L   = reinterpret_cast<float2*>(smem16);
    float*    P   = reinterpret_cast<float*>(smem16 + kOffP);
    float*    Dre = reinterpret_cast<float*>(smem16 + kOffD);
    float*    Dim = Dre + kDPad16;
    float*    hl  = reinterpret_cast<float*>(smem16 + kOffH);
    float2*   el  = reinterpret_cast<float2*>(smem16 + kOffE);
    float2*   Tl  = reinterpret_cast<float2*>(smem16 + kOffT);
    unsigned* cnt = reinterpret_cast<unsigned*>(smem16 + kOffC);

    const int t0   = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t0 >> 6), lane0 = t0 & 63;
    const int sigma = wave >> 3; // cross pass: this wave produces the radix-16 outputs q = 2 q' + sigma
    const float sg = sigma ? -1.f : 1.f;

    // ---- kernel-lifetime registers (exact table values)
#if GR4_C16_SWAP
    float2 twX[8], tw1[8], tw2[4], twE[4], Hr[8];
#else
    float2 twX[8], tw1[8], tw2[8], twE[4], Hr[8];
#endif
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        twX[k] = a.twX[t0 * 8 + k];
        tw1[k] = a.tw1[lane0 * 8 + k];
#if GR4_C16_SWAP
        if (k < 4) tw2[k] = a.tw2[(lane0 & 15) * 8 + k]; // W_64^{l' k}, l' = lane & 15
#else
        tw2[k] = a.tw2[(lane0 & 7) * 8 + k];
#endif
        Hr[k]  = FFTONLY ? make_float2(1.f, 0.f) : a.Hq[(wave * 64 + lane0) * 8 + k];
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) twE[r] = FFTONLY ? make_float2(1.f, 0.f) : a.twE[(wave * 64 + lane0) * 4 + r];
    if constexpr (!FFTONLY) {
        for (int i = t0; i < 2 * kDPad16; i += kT16) Dre[i] = 0.f;
        if (t0 < 272) hl[t0] = t0 < 256 ? a.taps[t0] : 0.f;
    }
    if (t0 < 4) cnt[t0] = 0u;
    __syncthreads(); // (no LDS-DMA in flight yet)

    // (s_setprio for the younger waves measured 2-5 % slower: the age-ordered arbitration keeps the waves apart, which is what overlaps their phases)
    const unsigned lds_L = __builtin_amdgcn_readfirstlane(lds_off(L)), lds_T = __builtin_amdgcn_readfirstlane(lds_off(Tl));
    long     f = blockIdx.x, fprev = -1;
    unsigned it = 0;
    if (f < a.n_frames) {
        if constexpr (!FFTONLY)
            if (wave < 2) dma16_1k((f > 0 ? a.x + f * kN16 - 256 : a.hist) + 128 * wave + 2 * lane0, lds_T + 1024u * wave);
        dma16_frame<0, 4>(a.x + f * kN16, lds_L, wave, lane0);
    }
    for (; f < a.n_frames; f += gridDim.x, ++it) {
        int t = threadIdx.x;
        asm volatile("" : "+v"(t)); // lane-dependent offsets are recomputed per frame instead of living in registers
        const int l = t & 63;
        float2*   R = reinterpret_cast<float2*>(smem16 + kOffW + wave * kRS); // this wave's private region
        G16_STAMP(0);
        G16_FULL_BARRIER(); // T: the frame has landed; every wave's |Y|^2 of the previous frame is in its region
        G16_STAMP(1);
        // ---- pick up the previous frame's |Y|^2: this thread stores bins t + 1024 j = 16 k' + q, q = t & 15, k' = (t >> 4) + 64 j
        float pend[8];
        {
            const float* src = reinterpret_cast<const float*>(smem16 + kOffW + (t & 15) * kRS) + (t >> 4);
#pragma unroll
            for (int j = 0; j < 8; ++j) pend[j] = src[64 * j];
        }
        // ---- phase 0: radix-16 across the frame's sixteen 512-sample blocks
        const int m = t & 511;
        float2    u[8];
        {
            const unsigned la = lds_L + 8u * (unsigned)m;
            f2v            d0[8], d1[8];
            G16_RD8(d0, la, 4096);
            G16_RD8(d1, la + 32768u, 4096);
            lds_wait8(d0);
            lds_wait8(d1);
            float2 hi7 = make_float2(d1[7][0], d1[7][1]);
#pragma unroll
            for (int h = 0; h < 8; ++h) u[h] = make_float2(fmaf(sg, d1[h][0], d0[h][0]), fmaf(sg, d1[h][1], d0[h][1]));
            if constexpr (!FFTONLY) {
                // x_f[N - 256 + j] is sample m + 512 * 15 of thread m = 256 + j:  Dz[j] = x_{f-1}[N - 256 + j] - x_f[N - 256 + j], j = 1..255
                if (sigma == 0 && m > 256) {
                    const int    j  = m - 256;
                    const float2 dd = csub(Tl[j], hi7);
                    Dre[j + (j >> 4)] = dd.x;
                    Dim[j + (j >> 4)] = dd.y;
                }
            }
            split_arrive(cnt + 0, l); // my reads of the private regions (pend), of the landing buffer and of the tail are issued
        }
        G16_STAMP(2);
        const rsrc_t rq = make_rsrc(a.out + (fprev < 0 ? 0 : fprev) * kN16, fprev < 0 ? 0u : (unsigned)(kN16 * sizeof(float)));
        // what leaves at which point of the frame: kSt[s] .. kSt[s + 1] = stores, kDm[s] .. kDm[s + 1] = DMA pieces issued at site s
        // sites: 0 after the cross reads, 1 after the cross butterflies (every wave has read the landing buffer: split barrier A), 2 after barrier #1,
        //        3 after stage 1, 4 after the X transform, 5 after the first E stage, 6 after the E transform.
        // Measured (tools/ab16.sh): any spread of the four pieces over sites 1..5 is within 1 %; all four at one site -3 %, a piece at site 6 -4 %
        // (it does not land before the next top barrier); stores anywhere in sites 0..5 within 1 %.
        constexpr int kSt[8] = {0, 2, 4, 6, 8, 8, 8, 8};
        constexpr int kDm[8] = {0, 0, 1, 2, 3, 4, 4, 4};
#define G16_SITE(sx)                                                                                   \
    do {                                                                                               \
        _Pragma("unroll") for (int j = kSt[sx]; j < kSt[(sx) + 1]; ++j) buf_store_f(rq, pend[j], t * 4, j * 4096); \
        if constexpr ((sx) >= 1) dma16_frame<kDm[sx], kDm[(sx) + 1]>(a.x + fn * kN16, lds_L, wave, l); \
    } while (0)
        const long fn = (f + gridDim.x < a.n_frames) ? f + gridDim.x : f; // last iteration re-reads its own frame: no divergent paths around the DMA
        G16_SITE(0);
        if (sigma) mul_w16_powers(u); // W_16^{h'} on the differences (wave-uniform branch)
        fft8(u); // u[q'] = Y_{2 q' + sigma}[m]
#pragma unroll
        for (int k = 0; k < 8; ++k) u[k] = cmul(u[k], twX[k]);
        G16_STAMP(3);
        split_wait(cnt + 0, 16u * (it + 1)); // every wave has picked up its share of the previous |Y|^2 and read the frame: regions, landing buffer and tail may be overwritten
        // ---- next frame into the landing buffer: in flight until the next top barrier, issue spread over the whole frame (smooth HBM demand)
        if constexpr (!FFTONLY)
            if (wave < 2) dma16_1k((fn > 0 ? a.x + fn * kN16 - 256 : a.hist) + 128 * wave + 2 * l, lds_T + 1024u * wave);
        G16_SITE(1);
#pragma unroll
        for (int k = 0; k < 8; ++k) *reinterpret_cast<float2*>(smem16 + kOffW + (2 * k + sigma) * kRS + 8 * m) = u[k];
        G16_STAMP(4);
        G16_LDS_BARRIER(); // #1: cross pass complete
        G16_STAMP(5);
        G16_SITE(2);

        // ---- e[n] = sum_j b[j] Dz[256 + n - j] on the MFMA units: waves 0..7, one (tile, K quarter) each, 16 MFMAs (see chain_fused.hip)
        if constexpr (!FFTONLY) {
            if (wave < 8) {
                using f32x4 = __attribute__((ext_vector_type(4))) float;
                const int    col = l & 15, kqm = l >> 4, kw = wave & 3;
                const float* pr  = ((wave >> 2) ? Dim : Dre) + 17 * col + kqm + 68 * kw;
                const float* pa  = hl + 256 + col - kqm - 64 * kw;
                f32x4        cr  = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int i0 = 0; i0 < 16; i0 += 4) {
                    float av[4], br[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        av[i] = pa[-4 * (i0 + i)];
                        br[i] = pr[4 * (i0 + i) + ((i0 + i) >> 2)];
                    }
#pragma unroll
                    for (int i = 0; i < 4; ++i) cr = __builtin_amdgcn_mfma_f32_16x16x4f32(av[i], br[i], cr, 0, 0, 0);
                }
                *reinterpret_cast<float4*>(P + (wave & 3) * 512 + (wave >> 2) * 256 + 16 * col + 4 * kqm) = make_float4(cr[0], cr[1], cr[2], cr[3]);
                split_arrive(cnt + 1, l);
            }
        }
        G16_FENCE();
        G16_STAMP(6);
        // ---- phase 1: this wave's 512-point transform of Y_q (q = wave)
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
        G16_STAMP(7);
        G16_SITE(3);
        private_tail(v, R, l, tw2);
        G16_STAMP(8);
        G16_SITE(4);
        float2 X[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) X[k] = FFTONLY ? v[k] : cmul(Hr[k], v[k]);
        G16_FENCE();
        if constexpr (!FFTONLY) {
            // ---- waves 0..3: e = sum of the four K-partial tiles (fixed order)
            if (wave < 4) { // the oldest waves win every issue arbitration and get here first: nobody waits for e
                split_wait(cnt + 1, 8u * (it + 1));
                const float* pp = P + t; // t < 256: e[t] = (re, im)
                el[t] = make_float2((pp[0] + pp[512]) + (pp[1024] + pp[1536]), (pp[256] + pp[768]) + (pp[1280] + pp[1792]));
                split_arrive(cnt + 2, l);
            }
            G16_SITE(5);
            // ---- phase 2: E[16 k' + q] = FFT_512( W_8192^{m q} e[m] ), only m < 256 non-zero: first stage on four inputs
            G16_STAMP(9);
            split_wait(cnt + 2, 4u * (it + 1));
            G16_STAMP(10);
            float2 e0 = cmul(el[l], twE[0]), e1 = cmul(el[64 + l], twE[1]), e2 = cmul(el[128 + l], twE[2]), e3 = cmul(el[192 + l], twE[3]);
            constexpr float hq = 0.70710678118654752440f;
            float2 o0 = e0, o1 = make_float2((e1.x + e1.y) * hq, (e1.y - e1.x) * hq), o2 = mul_mi(e2), o3 = make_float2((e3.y - e3.x) * hq, (-e3.x - e3.y) * hq);
            fft4(e0, e1, e2, e3); // even outputs
            fft4(o0, o1, o2, o3); // odd outputs
            v[0] = e0; v[2] = e1; v[4] = e2; v[6] = e3;
            v[1] = o0; v[3] = o1; v[5] = o2; v[7] = o3;
#pragma unroll
            for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], tw1[k]);
            G16_STAMP(11);
            private_tail(v, R, l, tw2);
            G16_STAMP(12);
            G16_SITE(6);
#pragma unroll
            for (int k = 0; k < 8; ++k) X[k] = cadd(X[k], v[k]);
        } else {
            G16_SITE(5);
            G16_SITE(6);
        }
        // ---- |Y[16 k' + q]|^2, k' = ka' + 8 kb0' + 64 kb1 on lane 8 ka' + kb0', into this wave's region (float index k'): picked up after the next top barrier
        {
            float* dst = reinterpret_cast<float*>(R) + c16_out_bin(l, 0);
#pragma unroll
            for (int k = 0; k < 8; ++k) dst[c16_out_bin(0, k)] = fmaf(X[k].x, X[k].x, X[k].y * X[k].y);
        }
        G16_STAMP(13);
        fprev = f;
    }
#undef G16_SITE
    if (fprev >= 0) { // the last frame's |Y|^2
        G16_FULL_BARRIER();
        const rsrc_t rq  = make_rsrc(a.out + fprev * kN16, kN16 * sizeof(float));
        const float* src = reinterpret_cast<const float*>(smem16 + kOffW + (t0 & 15) * kRS) + (t0 >> 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) buf_store_f(rq, src[64 * j], t0 * 4, j * 4096);
    }
}

// ------------------------------------------------------------------------------------------------ host side
struct Chain16 {
    DeviceBuffer d_twX, d_tw1, d_tw2, d_twE, d_Hq, d_taps;
};

template <typename T>
static int upload16(DeviceBuffer& b, const std::vector<T>& h) {
    int rc = b.ensure(h.size() * sizeof(T));
    if (rc) return rc;
    GR4_HIP_TRY(upload_fresh(b.ptr, h.data(), h.size() * sizeof(T)));
    return GR4HIP_OK;
}

static void put_w(std::vector<float>& v, size_t idx, long num, long den) { // W_den^num = exp(-2 pi i num / den), exact angle reduction
    const double ang = -2.0 * M_PI * (double)(((num % den) + den) % den) / (double)den;
    v[2 * idx]     = (float)std::cos(ang);
    v[2 * idx + 1] = (float)std::sin(ang);
}

// H: 8192 complex (interleaved floats) = FFT_8192 of the zero-padded taps; taps256: 256 floats
int chain16_create(Chain16** out, const float* H, const float* taps256) {
    auto* c = new (std::nothrow) Chain16();
    GR4_REQUIRE(c, "out of host memory");
    std::vector<float> twX(2 * 1024 * 8), tw1(2 * 64 * 8), tw2(2 * 16 * 8), twE(2 * 16 * 64 * 4), Hq(2 * 16 * 64 * 8), taps(taps256, taps256 + 256);
    for (int t = 0; t < 1024; ++t)
        for (int k = 0; k < 8; ++k) put_w(twX, (size_t)t * 8 + k, (long)(t & 511) * (2 * k + (t >> 9)), 8192);
    for (int l = 0; l < 64; ++l)
        for (int k = 0; k < 8; ++k) put_w(tw1, (size_t)l * 8 + k, (long)l * k, 512);
    for (int l0 = 0; l0 < 16; ++l0) // 8-point second stage: rows 0..7 (W_64^{l0 k}); swap variant: W_64^{l' k}, l' = 0..15, k = 0..3
        for (int k = 0; k < 8; ++k) put_w(tw2, (size_t)l0 * 8 + k, (long)l0 * k, 64);
    for (int q = 0; q < 16; ++q)
        for (int l = 0; l < 64; ++l) {
            for (int r = 0; r < 4; ++r) put_w(twE, ((size_t)q * 64 + l) * 4 + r, (long)(64 * r + l) * q, 8192);
            for (int kb1 = 0; kb1 < 8; ++kb1) {
                const int    bin = 16 * c16_out_bin(l, kb1) + q;
                const size_t i   = ((size_t)q * 64 + l) * 8 + kb1;
                Hq[2 * i]     = H[2 * bin];
                Hq[2 * i + 1] = H[2 * bin + 1];
            }
        }
    int rc = upload16(c->d_twX, twX);
    if (!rc) rc = upload16(c->d_tw1, tw1);
    if (!rc) rc = upload16(c->d_tw2, tw2);
    if (!rc) rc = upload16(c->d_twE, twE);
    if (!rc) rc = upload16(c->d_Hq, Hq);
    if (!rc) rc = upload16(c->d_taps, taps);
    if (rc) { delete c; return rc; }
    *out = c;
    return GR4HIP_OK;
}

void chain16_destroy(Chain16* c) { delete c; }
#ifdef GR4_C16_TIMING
static unsigned long long* g_dbg16 = nullptr;
#endif

int chain16_run(Chain16* c, const float* d_in, const float* d_hist256, size_t n_frames, float* d_out, unsigned max_wg, bool fft_only, hipStream_t st) {
    Chain16Args a{};
    a.x        = reinterpret_cast<const float2*>(d_in);
    a.hist     = reinterpret_cast<const float2*>(d_hist256);
    a.twX      = static_cast<const float2*>(c->d_twX.ptr);
    a.tw1      = static_cast<const float2*>(c->d_tw1.ptr);
    a.tw2      = static_cast<const float2*>(c->d_tw2.ptr);
    a.twE      = static_cast<const float2*>(c->d_twE.ptr);
    a.Hq       = static_cast<const float2*>(c->d_Hq.ptr);
    a.taps     = static_cast<const float*>(c->d_taps.ptr);
    a.out      = d_out;
    a.n_frames = (long)n_frames;
    a.dbg      = nullptr;
#ifdef GR4_C16_TIMING
    if (!g_dbg16) GR4_HIP_TRY(hipMalloc(&g_dbg16, (size_t)1 << 26));
    if (n_frames * 16 * 16 * 8 <= ((size_t)1 << 26)) a.dbg = g_dbg16;
#endif
    static PerDevice per_device;
    bool             first = false;
    int              dev = -1, n_cu = per_device.current(&first, &dev);
    GR4_REQUIRE(n_cu != 0, "chain16: cannot query the current device");
    if (first) {
        n_cu = -n_cu;
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain16_kernel<k16Mag2>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds16));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain16_kernel<k16FftMag2>), hipFuncAttributeMaxDynamicSharedMemorySize, kLds16));
        per_device.done(dev, n_cu);
    }
    const size_t   wgs  = max_wg ? std::min<size_t>(max_wg, (size_t)n_cu) : (size_t)n_cu;
    const unsigned grid = (unsigned)std::min<size_t>(n_frames, wgs);
    if (fft_only) hipLaunchKernelGGL(chain16_kernel<k16FftMag2>, dim3(grid), dim3(kT16), kLds16, st, a);
    else hipLaunchKernelGGL(chain16_kernel<k16Mag2>, dim3(grid), dim3(kT16), kLds16, st, a);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

} // namespace gr4

#ifdef GR4_C16_TIMING
extern "C" int gr4hip_dbg_c16_timing(unsigned long long* h_out, size_t n_frames) { // developer-only, not part of the ABI
    if (!gr4::g_dbg16) return GR4HIP_ERROR;
    gr4::hip_quiet(hipDeviceSynchronize());
    return hipMemcpy(h_out, gr4::g_dbg16, n_frames * 16 * 16 * 8, hipMemcpyDeviceToHost) == hipSuccess ? GR4HIP_OK : GR4HIP_RUNTIME_ERROR;
}
#endif
