// chain_td.hip -- fused complex<float> FIR -> fftSize-point FFT -> |X|^2 with the filter in the TIME domain (GR4HIP_CHAIN_FUSED_TD), fftSize 256 .. 4096.
//
// The same runtime fusion as chain_fused.hip -- fir_filter (blocks/filter/.../time_domain_filter.hpp:44-47) + FFT block (blocks/fourier/.../fft.hpp:147-171)
// + mag2, the analogue of Merge<fir,"out",fft,"in"> (core/include/gnuradio-4.0/BlockMerging.hpp:136-320) -- but with the reference's own arithmetic for the
// filter: the direct-form sum, as the block-Toeplitz contraction of fir_batched.hip -- on the bf16 matrix pipe with three-term splits of samples and taps
// (fir_bf16.hip: float32 accuracy; the first version used the f32 MFMA, whose measurements the notes below refer to).  Two reasons to have it beside the fast
// convolution:
//  * short filters.  At <= 64 taps the direct form costs 320 executed flop per complex sample on the MATRIX pipe, which the frame transform (VALU) does not
//    use: one workgroup's MFMA phase runs beside another workgroup's transform, and the chain needs ONE transform per frame instead of the fast
//    convolution's three (forward, inverse, windowed forward).
//  * dynamic range.  The fast convolution carries the rounding of its transforms (~2e-6 of the INPUT rms); this kernel has the error of a
//    float32 dot product relative to the OUTPUT.  (The guard of chain.hip sent its streams here until late round 3; under a rejected signal 50 dB above the output the
//    three-term bf16 products measure 3 .. 16 x a float32 sum's error, so the guard's destination is now the kernel pair with float32 products, chain.hip.)
//
// One workgroup (256 lanes) per segment of 4096 complex samples = 4096 / fftSize frames:
//   stage    the segment + Kp samples in front of it, de-interleaved into a re and an im plane (18 / 16-padded, planes 16 banks apart) -- requested into
//            registers at once (17 loads in flight per lane)
//   MFMA     wave w: tiles of 256 outputs, re and im accumulators under the same A operand (Toeplitz: read from the tap row in LDS); D x window -> the frame
//            buffer in LDS (natural order, one pad float2 per 32)
//   FFT      16 points per lane, fftSize / 16 lanes per frame: Stockham 16 x 16 x (fftSize / 256), the plan of fft_fast_kernel with its first pass fed
//            from LDS instead of HBM; |X|^2 straight from registers (lane t holds bins t + j fftSize / 16: coalesced)
// The frame buffer takes the place of the staged samples: LDS 38 KB -> four workgroups per CU, whose phases interleave on the two pipes.  HBM traffic: 8 B in + 4 B out per sample.  Bound: MFMA f32 (2 * 2 * (Kp + 16) flop per sample).
#ifndef GR4_TD_LOAD_AUX // cache policy of the sample loads (2 = nt) / the spectrum stores: developer builds override; profiles/r05_streaming_hints.txt
#define GR4_TD_LOAD_AUX 0
#endif
#ifndef GR4_TD_STORE_NT
#define GR4_TD_STORE_NT 0
#endif
#include "common.hpp"
#include "buffer_ops.hpp"
#include "fft_radix.hpp"
#include "fir_f16_common.hpp" // (hf_wave_sum)

#include <cmath>

namespace gr4 {

void fir_bf16_make_afrag(const float* taps, size_t ntaps, int* KS_out, std::vector<unsigned short>* af, size_t nch, int force_ks); // fir_bf16.hip
int  make_window(int window, float* w, size_t n, float beta);                                                               // runtime.hip

using td_f32x4 = __attribute__((ext_vector_type(4))) float;
#ifndef GR4_TD_EIGHT_TERMS
#define GR4_TD_EIGHT_TERMS 1
#endif
using td_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
using td_bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using td_f32x2  = __attribute__((ext_vector_type(2))) float;
using td_u32x4  = __attribute__((ext_vector_type(4))) unsigned;
// two samples -> their three bf16 terms, each pair packed in one dword (fir_bf16.hip: x = h + m + l to 2^-25 |x|, residuals exact in float32)
__device__ __forceinline__ void td_split2(float x0, float x1, unsigned& h, unsigned& m, unsigned& l) {
    const td_f32x2  v  = {x0, x1};
    const td_bf16x2 hh = __builtin_convertvector(v, td_bf16x2);
    const td_f32x2  r1 = v - __builtin_convertvector(hh, td_f32x2);
    const td_bf16x2 mm = __builtin_convertvector(r1, td_bf16x2);
    const td_f32x2  r2 = r1 - __builtin_convertvector(mm, td_f32x2);
    const td_bf16x2 ll = __builtin_convertvector(r2, td_bf16x2);
    h = __builtin_bit_cast(unsigned, hh);
    m = __builtin_bit_cast(unsigned, mm);
    l = __builtin_bit_cast(unsigned, ll);
}
// every barrier of the kernel orders LDS accesses only.  __syncthreads() also waits for the global stores in flight (vmcnt(0)): the |X|^2 stores of a
// segment would have to land before its workgroup may stage the next one -- measured 245 instead of 3xx Gsamples/s at 64 taps
#define GR4_TD_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
constexpr int kTdSeg = 4096; // complex samples per segment

// waves per SIMD the kernel is compiled for (512 / W registers per lane)
template <int KS, int LOG2N>
constexpr int td_waves() {
#ifdef GR4_TD_W
    return GR4_TD_W;
#else
    return 2; // (eight accumulators, twelve B operands and the tap fragments of a K-step want ~180 .. 230 registers; 2 / 3 / 4 waves per SIMD measured alike on the f32 form)
#endif
}

template <int KS /*K-steps of 32: window 32 KS, Hb = 32 KS - 16 samples in front of a block*/, int LOG2N>
__global__ __launch_bounds__(256, (td_waves<KS, LOG2N>())) void chain_td_kernel(const float2* __restrict__ x, const float2* __restrict__ hist /*the Kh samples in front of x*/, int Kh,
                                                        const td_u32x4* __restrict__ afrag /*[3 planes][KS][64 lanes]: 8 bf16 each (fir_bf16_make_afrag)*/,
                                                        const float* __restrict__ win /*[N] or null*/, const float2* __restrict__ tw /*W_N^j*/, float* __restrict__ out,
                                                        long n /*samples: whole frames*/, float2* __restrict__ new_hist,
                                                        float gthr /*> 0: the FIR guard of fir.hip -- a segment whose FILTER output carries less than gthr x the power of its samples is marked*/,
                                                        unsigned char* __restrict__ flags /*one byte per 4096-sample segment: chain_redo_kernel (chain_fused.hip) evaluates the marked ones again*/) {
    constexpr int Hb = 32 * KS - 16, NS = kTdSeg + Hb, N = 1 << LOG2N, T = N / 16, NP = N + N / 32;
    constexpr int PL  = NS + 8;                         // bf16 elements per plane
    constexpr int NL4 = (NS / 2 + 255) / 256;          // float4 loads (two complex samples each) per lane and segment
    constexpr int R3  = N / 256;                       // third pass radix (1: none)
    constexpr int B3  = R3 > 1 ? 16 / R3 : 1, NB3 = N / (R3 > 1 ? R3 : 1);
    extern __shared__ __attribute__((aligned(16))) float td_sm[];
    __shared__ float jred[8]; // the judge's wave sums: input power, output power
    unsigned short* pl = reinterpret_cast<unsigned short*>(td_sm); // [6][PL] staged samples: re h, m, l, im h, m, l ...
    float2*         fb = reinterpret_cast<float2*>(td_sm);         // ... then, in the same place, the segment's frames [4096 / N][NP]
    auto    P  = [](int i) { return i + (i >> 5); };
    auto    PB = [](int s_) { return s_; }; // (no padding: see fir_bf16.hip)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int fl = tid / T, t = tid % T; // FFT phase: frame slot, lane of the frame

    auto put2 = [&](int q, float4 v) { // complex samples 2 q, 2 q + 1 of the staged range -> the six planes
        unsigned h, m, l;
        const int e = PB(2 * q);
        td_split2(v.x, v.z, h, m, l);
        *reinterpret_cast<unsigned*>(pl + e)          = h;
        *reinterpret_cast<unsigned*>(pl + PL + e)     = m;
        *reinterpret_cast<unsigned*>(pl + 2 * PL + e) = l;
        td_split2(v.y, v.w, h, m, l);
        *reinterpret_cast<unsigned*>(pl + 3 * PL + e) = h;
        *reinterpret_cast<unsigned*>(pl + 4 * PL + e) = m;
        *reinterpret_cast<unsigned*>(pl + 5 * PL + e) = l;
    };
    // per-lane twiddle bases, exact table values
    const float2 w2a_ = tw[(t & 15) * (N / 256)], w2b_ = tw[2 * (t & 15) * (N / 256)]; // W_256^k, W_256^2k
    float2       w3_ = make_float2(1.f, 0.f), w3sq_ = make_float2(1.f, 0.f);
    if constexpr (R3 > 1) w3_ = tw[t & 255];         // W_N^t  (256 R3 = N)
    if constexpr (R3 == 16) w3sq_ = tw[2 * (t & 255)];

    // Persistent grid: as many workgroups per CU as the kernel is compiled for, segments dealt round-robin.  (Tried: a different wave priority for each of
    // the workgroups that share a CU, so that they would not fall into step on the matrix pipe -- no change, 245 Gsamples/s with and without.)
    const long nseg = (n + kTdSeg - 1) / kTdSeg;
    const long n_frames = n >> LOG2N;
    for (long sg = blockIdx.x; sg < nseg; sg += gridDim.x) {
        const long seg0 = sg * kTdSeg;
        float      pxl  = 0.f; // this lane's share of the segment's input power (its own 4096 samples, not the history in front of them)
        if (sg > 0) { // (no register prefetch of the next segment: with several workgroups per CU another one always has work)
            const long   i0   = seg0 - Hb; // seg0 >= kTdSeg > Hb: nothing below 0; past the end of the span / of the segment the range check returns 0
            const long   nrec = n - i0 < (long)NS ? n - i0 : (long)NS;
            const rsrc_t r    = make_rsrc(x + i0, (unsigned)(nrec > 0 ? nrec * 8 : 0));
            float4       nxt[NL4];
#pragma unroll
            for (int u = 0; u < NL4; ++u) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, 256 * u * 16, GR4_TD_LOAD_AUX);
                nxt[u]       = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            }
#pragma unroll
            for (int u = 0; u < NL4; ++u) {
                const int q = tid + 256 * u;
                if (q < NS / 2) {
                    put2(q, nxt[u]);
                    if (2 * q >= Hb) pxl = fmaf(nxt[u].x, nxt[u].x, fmaf(nxt[u].y, nxt[u].y, fmaf(nxt[u].z, nxt[u].z, fmaf(nxt[u].w, nxt[u].w, pxl)))); // (Hb is even: a pair never straddles the segment's start)
                }
            }
        } else {
            for (int q = tid; q < NS / 2; q += 256) { // the first segment of the span reads the carried history in front of x
                float2 t2[2];
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const long i = 2L * q + c - Hb;
                    t2[c]        = i >= 0 ? (i < n ? x[i] : make_float2(0.f, 0.f)) : (i >= -(long)Kh ? hist[Kh + i] : make_float2(0.f, 0.f));
                }
                put2(q, make_float4(t2[0].x, t2[0].y, t2[1].x, t2[1].y));
                if (2 * q >= Hb) pxl = fmaf(t2[0].x, t2[0].x, fmaf(t2[0].y, t2[0].y, fmaf(t2[1].x, t2[1].x, fmaf(t2[1].y, t2[1].y, pxl))));
            }
        }
        GR4_TD_BARRIER();

        float4 wq[4]; // this lane's window values, requested before the MFMAs (the frame-buffer writes behind them would otherwise wait for the L2)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pos = (16 * (16 * (4 * wave + q) + col) + 4 * kq) & (N - 1);
            wq[q]         = win ? *reinterpret_cast<const float4*>(win + pos) : make_float4(1.f, 1.f, 1.f, 1.f);
        }
        // ---- direct-form FIR on the bf16 matrix pipe with three-term splits (fir_bf16.hip): this wave's four tiles of 256 outputs, re and im under the same A
        td_f32x4 acr[4], aci[4];
        const td_u32x4* afl = afrag + lane; // (laundered: the loads are loop-invariant and would be hoisted out of the segment loop, registers and all)
        asm volatile("" : "+v"(afl));
        td_u32x4 a[3][KS]; // A fragments of the three tap planes: fetched (L2) per segment, so that they do not occupy registers through the transform phase
#pragma unroll
        for (int p = 0; p < 3; ++p)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a[p][ks] = afl[(p * KS + ks) * 64];
#pragma unroll
        for (int pp = 0; pp < 2; ++pp) { // two tiles (eight accumulators) at a time
            const int ib0 = 16 * (4 * wave + 2 * pp), ib1 = ib0 + 16; // first 16-sample block of each tile
            td_f32x4  cr0 = {0.f, 0.f, 0.f, 0.f}, dr0 = cr0, ci0 = cr0, di0 = cr0, cr1 = cr0, dr1 = cr0, ci1 = cr0, di1 = cr0;
            const int s0 = 16 * (ib0 + col) + 8 * kq, s1 = 16 * (ib1 + col) + 8 * kq;
#ifndef GR4_TD_NO_MFMA // (developer builds: one phase removed, tools/build_variant.sh)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                const unsigned short* q0 = pl + PB(s0 + 32 * ks);
                const unsigned short* q1 = pl + PB(s1 + 32 * ks);
                const td_bf16x8 ah = __builtin_bit_cast(td_bf16x8, a[0][ks]), am = __builtin_bit_cast(td_bf16x8, a[1][ks]), al = __builtin_bit_cast(td_bf16x8, a[2][ks]);
                auto six = [&](const unsigned short* qp, td_f32x4& c, td_f32x4& d) { // hh, hm, mh -> c; hl, lh, mm -> d
                    const td_bf16x8 bh = *reinterpret_cast<const td_bf16x8*>(qp), bm = *reinterpret_cast<const td_bf16x8*>(qp + PL), bl = *reinterpret_cast<const td_bf16x8*>(qp + 2 * PL);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, c, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, d, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, c, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, d, 0, 0, 0);
                    c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, c, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, d, 0, 0, 0);
#if GR4_TD_EIGHT_TERMS // + ml, lm: the products are then exact to ~2^-31 -- below a float32 product's own 2^-25.  This chain has no dynamic-range guard (AUTO takes it for
                    //   <= 64 taps): under a rejected interferer 50 dB above the output the six-term form measured 3 x the float32 CPU sum's error (profiles/r03_fuzz_summary.txt)
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bl, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bm, d, 0, 0, 0);
#endif
                };
                six(q0, cr0, dr0);
                six(q0 + 3 * PL, ci0, di0);
                six(q1, cr1, dr1);
                six(q1 + 3 * PL, ci1, di1);
            }
#else
            cr0[0] = (float)pl[PB(s0)]; ci0[0] = (float)pl[3 * PL + PB(s0)]; cr1[0] = (float)pl[PB(s1)]; ci1[0] = (float)pl[3 * PL + PB(s1)];
#endif
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acr[2 * pp][r] = cr0[r] + dr0[r]; aci[2 * pp][r] = ci0[r] + di0[r];
                acr[2 * pp + 1][r] = cr1[r] + dr1[r]; aci[2 * pp + 1][r] = ci1[r] + di1[r];
            }
        }
        if (gthr > 0.f) { // the filter's output power (in front of the window), the quietest wave's counting for the segment like in the band kernels
            float pyl = 0.f;
#pragma unroll
            for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int r = 0; r < 4; ++r) pyl = fmaf(acr[q][r], acr[q][r], fmaf(aci[q][r], aci[q][r], pyl)); // (outputs past the span's end: the filter's ringing over zeros -- a few taps' worth in the last segment)
            pxl = hf_wave_sum(pxl);
            pyl = hf_wave_sum(pyl);
            if (lane == 0) { jred[wave] = pxl; jred[4 + wave] = seg0 + 1024L * wave < n ? pyl : __builtin_inff(); } // (a wave past the end of the span: nothing to judge)
        }
        GR4_TD_BARRIER(); // every wave is done with the staged samples: the frame buffer takes their place
        if (gthr > 0.f && tid == 0) {
            const float px = (jred[0] + jred[1]) + (jred[2] + jred[3]);
            const float py = 4.f * __builtin_fminf(__builtin_fminf(jred[4], jred[5]), __builtin_fminf(jred[6], jred[7]));
            // two verdicts: the quietest wave's quarter at the FIR guard's 21 dB, the whole segment 6 dB earlier (this kernel squares the filter output behind its transform, which doubles
            // the split products' relative error -- chain.hip kChainPairGuardRatio; the quietest-part statistic at 15 dB dips below on narrow-band noise alone)
            const float pa = (jred[4] + jred[5]) + (jred[6] + jred[7]);
            flags[sg] = (py < gthr * px || pa < 4.f * gthr * px) ? 3 : 0; // (a NaN power compares false: unmarked)
        }
        // D[row = 4 kq + r][col] of tile q: sample 16 (16 (4 wave + q) + col) + 4 kq + r of the segment; x window -> frame buffer (natural order)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = 16 * (16 * (4 * wave + q) + col) + 4 * kq, f = o >> LOG2N, pos = o & (N - 1);
            const float4 w = wq[q];
            float2* d = fb + f * NP + P(pos); // pos % 4 == 0: the four samples share a 32-block
            d[0]      = make_float2(acr[q][0] * w.x, aci[q][0] * w.x);
            d[1]      = make_float2(acr[q][1] * w.y, aci[q][1] * w.y);
            d[2]      = make_float2(acr[q][2] * w.z, aci[q][2] * w.z);
            d[3]      = make_float2(acr[q][3] * w.w, aci[q][3] * w.w);
        }
        GR4_TD_BARRIER();

        // ---- the segment's frames: 16 x 16 x R3
        float2* buf = fb + fl * NP;
        float2  v[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = buf[P(t + r * T)];
#ifdef GR4_TD_NO_FFT
        {
            const long frame = (seg0 >> LOG2N) + fl;
            if (frame < n_frames)
                for (int j = 0; j < 16; ++j) {
#if GR4_TD_STORE_NT
                    __builtin_nontemporal_store(fmaf(v[j].x, v[j].x, v[j].y * v[j].y), &out[frame * N + t + j * T]);
#else
                    out[frame * N + t + j * T] = fmaf(v[j].x, v[j].x, v[j].y * v[j].y);
#endif
                }
            GR4_TD_BARRIER(); continue;
        }
#endif
        GR4_TD_BARRIER();
        fft16<1>(v);
#pragma unroll
        for (int r = 0; r < 16; ++r) buf[P(16 * t + r)] = v[perm16(r)];
        GR4_TD_BARRIER();
        // pass 2 (p = 16, radix 16): butterfly i = t, k = t & 15
#pragma unroll
        for (int r = 0; r < 16; ++r) v[r] = buf[P(t + r * T)];
        float2 w2a = w2a_, w2b = w2b_, w3 = w3_, w3sq = w3sq_; // (laundered: their power chains are loop-invariant and would be kept as ~35 registers)
        asm volatile("" : "+v"(w2a.x), "+v"(w2a.y), "+v"(w2b.x), "+v"(w2b.y), "+v"(w3.x), "+v"(w3.y), "+v"(w3sq.x), "+v"(w3sq.y));
        apply_powers(v, w2a, w2b);
        fft16<1>(v);
        float2 X[16]; // X[j] = bin t + j T
        if constexpr (R3 == 1) {
#pragma unroll
            for (int q = 0; q < 16; ++q) X[q] = v[perm16(q)];
        } else {
            GR4_TD_BARRIER();
#pragma unroll
            for (int q = 0; q < 16; ++q) buf[P((t & ~15) * 16 + (t & 15) + 16 * q)] = v[perm16(q)];
            GR4_TD_BARRIER();
            // pass 3 (p = 256, radix R3): butterflies i_b = t + b T, k = i_b & 255
#pragma unroll
            for (int b = 0; b < B3; ++b)
#pragma unroll
                for (int r = 0; r < R3; ++r) v[b * R3 + r] = buf[P(t + b * T + r * NB3)];
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
#pragma unroll
            for (int b = 0; b < B3; ++b)
#pragma unroll
                for (int q = 0; q < R3; ++q) X[b + q * B3] = v[b * R3 + (R3 == 16 ? perm16(q) : q)];
        }
        const long frame = (seg0 >> LOG2N) + fl;
        if (frame < n_frames) {
            float* o = out + frame * N + t;
#pragma unroll
            for (int j = 0; j < 16; ++j) o[j * T] = fmaf(X[j].x, X[j].x, X[j].y * X[j].y);
        }
        GR4_TD_BARRIER(); // every lane is done with the frame buffer before the next segment is staged over it
    }
    if (new_hist != nullptr && blockIdx.x == 0) { // the other half of the caller's ping-pong pair: nobody reads it in this launch
        for (int h = tid; h < Kh; h += 256) {
            const long i = n - Kh + h;
            new_hist[h]  = i >= 0 ? x[i] : hist[Kh + i];
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------------
// Why the kernel above alternates and does not pipeline.  Two pipelined forms were built and measured against it (64 taps -> 1024-pt Hann, 251 Gsamples/s):
//  * 512 lanes, waves 0..3 multiply segment i while waves 4..7 transform segment i - 1, six workgroup barriers per segment: 177;
//  * producer / consumer wave pairs (wave w multiplies blocks of 1024 samples, wave w + 4 transforms them; frame slots and two LDS counters per pair, no
//    workgroup barrier at all, MFMA / LDS-read order pinned with sched_group_barrier): 230.
// In every form the time is the SUM of the filter's and the transform's: the M role alone runs the matrix pipe at 84 % (396 Gsamples/s), the F role alone
// at 298 .. 438, both together never beat the alternating kernel; counters of the latter: v_mfma_f32 busy 53 % + other VALU 29 %.  The f32 MFMA has the
// FP32 VALU's rate (64 flop / clk / SIMD) and, by these measurements, its issue slot: a SIMD does one or the other.  "Matrix pipe beside the VALU" holds
// for the reduced-precision MFMAs, not for this one -- so 2.33 ps (filter) + ~0.7 ps (transform) + staging per sample bound this chain at ~300 Gsamples/s.
// (Also measured: pinning the order in the alternating kernel changes nothing -- other waves hide the operand latency -- and the solver behind
// sched_group_barrier does not finish on a 272-MFMA region.)

struct ChainTd {
    float        gthr = 0.f;      // (sum b^2) / 128: the FIR guard (fir.hip kGuardSegmentRatio); the kernel judges the whole segment at 4 x this as well; chain_td_process passes it unless the guard is off
    DeviceBuffer d_flags;         // one byte per 4096-sample segment of the last launch
    size_t       ntaps = 0, N = 0;
    int          KS = 0, Kp = 0, log2n = 0;
    bool         windowed = false;
    DeviceBuffer d_afrag, d_win, d_tw, d_hist[2];
    int          cur = 0, dev = 0;
    bool         zero_hist = true; // reset asked for (or nothing has run yet): d_hist[cur] is zeroed on the stream of the next call that reads it (common.hpp, the stream rule)
};

int chain_td_supported(size_t ntaps, size_t fft_size, int window) {
    return is_pow2(fft_size) && fft_size >= 256 && fft_size <= 4096 && ntaps >= 1 && ntaps <= 256 && window >= GR4HIP_WIN_NONE && window <= GR4HIP_WIN_KAISER;
}

int chain_td_reset(ChainTd* c) { // host-side note only: a launch still in flight may be writing d_hist[cur] (its `new_hist`); the zeroing goes behind it on the next call's stream
    c->zero_hist = true;
    return GR4HIP_OK;
}
static int td_history_on(ChainTd* c, hipStream_t st) {
    if (c->zero_hist) {
        GR4_HIP_TRY(hipMemsetAsync(c->d_hist[c->cur].ptr, 0, (size_t)c->Kp * sizeof(float2), st));
        c->zero_hist = false;
    }
    return GR4HIP_OK;
}

int chain_td_create(ChainTd** out, const float* taps, size_t ntaps, size_t fft_size, int window) {
    if (!chain_td_supported(ntaps, fft_size, window)) { set_error("fused time-domain chain: unsupported configuration (fft_size 256 .. 4096, <= 256 taps)"); return GR4HIP_UNSUPPORTED; }
    auto* c = new (std::nothrow) ChainTd();
    GR4_REQUIRE(c, "out of host memory");
    c->ntaps = ntaps;
    c->N     = fft_size;
    {
        double h2 = 0;
        for (size_t k = 0; k < ntaps; ++k) h2 += (double)taps[k] * taps[k];
        c->gthr = (float)(h2 / 128.0); // (the kernel's second verdict is at 4 x this -- see there; until the round's last day this was h2 / 32 on the quietest-wave statistic alone) // (the FIR guard of fir.hip is at 1 / 128: this kernel squares the filter output behind its transform, which doubles the split products' relative error -- chain.hip kChainPairGuardRatio)
    }
    (void)hipGetDevice(&c->dev);
    c->log2n = (int)ilog2(fft_size);
    std::vector<unsigned short> af;
    c->Kp = ntaps <= 64 ? 64 : ntaps <= 128 ? 128 : 256;             // carried history: Kp complex samples
    c->KS = ntaps <= 81 ? 3 : ntaps <= 145 ? 5 : 9;                   // window of 32 KS samples, Hb = 32 KS - 16 >= taps - 1
    fir_bf16_make_afrag(taps, ntaps, &c->KS, &af, 1, c->KS);
    auto up = [](DeviceBuffer& b, const void* p, size_t bytes) -> int {
        int rc = b.ensure(bytes);
        if (rc) return rc;
        GR4_HIP_TRY(upload_fresh(b.ptr, p, bytes));
        return GR4HIP_OK;
    };
    int rc = up(c->d_afrag, af.data(), af.size() * sizeof(unsigned short));
    c->windowed = window != GR4HIP_WIN_NONE && window != GR4HIP_WIN_RECTANGULAR;
    if (!rc && c->windowed) {
        std::vector<float> w(fft_size);
        rc = make_window(window, w.data(), fft_size, 1.6f); // fft.hpp:141: default beta
        if (!rc) rc = up(c->d_win, w.data(), w.size() * sizeof(float));
    }
    if (!rc) {
        std::vector<float> ts(2 * fft_size);
        for (size_t k = 0; k < fft_size; ++k) {
            const double ang = -2.0 * M_PI * (double)k / (double)fft_size;
            ts[2 * k]     = (float)std::cos(ang);
            ts[2 * k + 1] = (float)std::sin(ang);
        }
        rc = up(c->d_tw, ts.data(), ts.size() * sizeof(float));
    }
    for (int k = 0; k < 2 && !rc; ++k) rc = c->d_hist[k].ensure((size_t)c->Kp * sizeof(float2));
    if (!rc) rc = chain_td_reset(c);
    if (rc) { delete c; return rc; }
    *out = c;
    return GR4HIP_OK;
}

void chain_td_destroy(ChainTd* c) { delete c; }

// the carried history: the last Kp complex samples of the stream, hist[h] = x[-Kp + h] of the next call; valid for work enqueued on `st` behind this call
// (null: a pending reset could not be enqueued)
const float* chain_td_history(ChainTd* c, int* Kp, hipStream_t st) {
    if (Kp) *Kp = c->Kp;
    return td_history_on(c, st) ? nullptr : static_cast<const float*>(c->d_hist[c->cur].ptr);
}
// start from the 256 complex samples in front of the next call (the fused frequency-domain kernel's convention)
int chain_td_set_history256(ChainTd* c, const float* d_hist256, hipStream_t st) {
    c->zero_hist = false;
    GR4_HIP_TRY(hipMemcpyAsync(c->d_hist[c->cur].ptr, d_hist256 + 2 * (256 - c->Kp), (size_t)c->Kp * sizeof(float2), hipMemcpyDeviceToDevice, st));
    return GR4HIP_OK;
}

template <int KS>
static int td_launch(ChainTd* c, const float* d_in, size_t n_frames, float* d_mag2, hipStream_t st, bool judged) {
    constexpr int NS = kTdSeg + 32 * KS - 16, PL = NS + 8;
    const size_t  lds = std::max((size_t)6 * PL * sizeof(unsigned short), (size_t)(kTdSeg + kTdSeg / 32) * sizeof(float2)); // six bf16 planes of staged samples, then the frames
    const long    n   = (long)(n_frames * c->N), nseg = ceil_div(n, (long)kTdSeg);
    int           n_cu = 0;
    GR4_HIP_TRY(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, c->dev));

    const auto    xc = reinterpret_cast<const float2*>(d_in);
    const auto    hc = static_cast<const float2*>(c->d_hist[c->cur].ptr);
    const auto    nh = static_cast<float2*>(c->d_hist[c->cur ^ 1].ptr);
    const auto    af = static_cast<const td_u32x4*>(c->d_afrag.ptr);
    const float*  wn = c->windowed ? static_cast<const float*>(c->d_win.ptr) : nullptr;
    const auto    tw = static_cast<const float2*>(c->d_tw.ptr);
    const float   gthr = judged && c->ntaps > 1 ? c->gthr : 0.f;
    unsigned char* fl  = nullptr;
    if (gthr > 0.f) {
        if (const int rc = c->d_flags.ensure((size_t)nseg)) return rc;
        fl = static_cast<unsigned char*>(c->d_flags.ptr);
    }
#define GR4_TD_CASE(L2)                                                                                                                    \
    case L2: {                                                                                                                             \
        auto kern = chain_td_kernel<KS, L2>;                                                                                               \
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));       \
        const dim3 grid((unsigned)std::min<long>(nseg, (long)n_cu * td_waves<KS, L2>()));                          \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, xc, hc, c->Kp, af, wn, tw, d_mag2, n, nh, gthr, fl);                       \
    } break
    switch (c->log2n) {
        GR4_TD_CASE(8);
        GR4_TD_CASE(9);
        GR4_TD_CASE(10);
        GR4_TD_CASE(11);
        GR4_TD_CASE(12);
    default: return GR4HIP_UNSUPPORTED;
    }
#undef GR4_TD_CASE
    GR4_LAUNCH_CHECK();
    c->cur ^= 1;
    return GR4HIP_OK;
}

// judged: every 4096-sample segment's filter output power is compared with its input power and the verdict left in chain_td_flags() -- one byte per segment; the caller
// (chain.hip) enqueues chain_fused_redo behind this launch with the history this call STARTED from (chain_td_history before the call)
int chain_td_process(ChainTd* c, const float* d_in, size_t n_frames, float* d_mag2, hipStream_t st, bool judged) {
    if (n_frames == 0) return GR4HIP_OK;
    GR4_REQUIRE((uintptr_t)d_in % 8 == 0 && (uintptr_t)d_mag2 % 4 == 0, "fused time-domain chain: misaligned device pointer");
    if (const int rc = td_history_on(c, st)) return rc;
    switch (c->KS) {
    case 3: return td_launch<3>(c, d_in, n_frames, d_mag2, st, judged);
    case 5: return td_launch<5>(c, d_in, n_frames, d_mag2, st, judged);
    default: return td_launch<9>(c, d_in, n_frames, d_mag2, st, judged);
    }
}
const unsigned char* chain_td_flags(const ChainTd* c) { return static_cast<const unsigned char*>(c->d_flags.ptr); }

} // namespace gr4
