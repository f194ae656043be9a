// chain_fused.hip -- fused complex<float> FIR -> 8192-pt FFT -> |X|^2 for gfx950 (GR4HIP_CHAIN_FUSED_FD).
//
// The runtime fusion of fir_filter (blocks/filter/.../time_domain_filter.hpp:44-47) with the FFT block
// (blocks/fourier/.../fft.hpp:147-171) and a mag2 epilogue: the analogue of Merge<fir,"out",fft,"in">
// (core/include/gnuradio-4.0/BlockMerging.hpp:136-320) -- one launch, no intermediate buffer in HBM.
//
// A direct-form 256-tap complex FIR costs 1024 flop per 8-byte sample and caps the chain at ~150 Gsamples/s
// (FP32-bound, 23 % of the 12 B/sample HBM roofline).  The frame spectrum of the FILTERED stream is instead
// obtained in the frequency domain.  With x_f the f-th frame (N = 8192 samples), X = FFT(x_f), H = FFT(taps):
//     FFT(y_f)[k] = H[k] * X[k] + E[k],      E = FFT_N(e),
//     e[n] = sum_{j>n} b[j] * d[255 + n - j]  (n < 255),   d[i] = x_{f-1}[N-255+i] - x_f[N-255+i]
// H*X is the circular convolution; e is the exact linear-vs-circular correction of the first K-1 outputs (the
// samples that must see the previous frame's tail instead of this frame's own tail).  e is 255 samples long, so its
// zero-padded transform skips the first Stockham pass (a 32-point DFT of one non-zero input is a broadcast).
// Cost: ~1.7 FFT equivalents (~140 flop/sample) instead of 1024 + 65 flop/sample; HBM traffic = 8 B in + 4 B out
// (+ 2 KB of the previous frame's tail per 64 KB frame).
//
// The same kernel body serves three outputs (template MODE): |FFT(y_f)|^2 with the rectangular window (the formula above), |FFT(w y_f)|^2
// for any window (y_f recovered by an inverse transform, + e, x w, transformed again), and y_f itself (fir_filter<complex<float>> as a
// fast convolution).
//
// One persistent workgroup of 512 lanes per CU, one frame at a time; Stockham radix 32 x 16 x 16, every lane holds 16 complex points
// (the radix-32 pass pairs neighbouring lanes through DPP).  The next frame streams into a second LDS buffer by LDS-DMA while the
// current one is transformed; e comes from the MFMA units.  Exchange layouts are padded (rows of 272 / 513 float2) so that all
// ds_read_b64 / ds_write_b64 are bank-conflict-free.  DESIGN.md 3.1 has the phase table, the measurements and what was tried.
// (round 5) the frame pipeline's HBM accesses are streaming: the samples come in once through the LDS-DMA and the results leave once -- both marked nt.  A/B on one box, alternated
// (profiles/r05_headline_bounds.txt): stores alone +0.5 %, loads alone +0.5 %, both +1.3 / +1.4 %.  Developer builds override the two macros on the command line.
#ifndef GR4_BUF_STORE_AUX
#define GR4_BUF_STORE_AUX 2
#endif
#ifndef GR4_DMA_MODS
#define GR4_DMA_MODS " nt"
#endif
#include "common.hpp"
#include "buffer_ops.hpp"
#include "fft_radix.hpp"
#include "fir_f16_common.hpp"

#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>

#ifndef GR4_E_BF16
#define GR4_E_BF16 1 // the correction e on the bf16 matrix pipe (three-term splits); 0: on v_mfma_f32_16x16x4_f32 as before
#endif
#ifndef GR4_FMA_COMBINE
#define GR4_FMA_COMBINE 1
#endif
#ifndef GR4_E_BF16_WIN
#define GR4_E_BF16_WIN 1 // the same in the windowed modes (their LDS image is 160 KiB to the byte with it)
#endif

namespace gr4 {

constexpr int kN     = 8192;
constexpr int kT     = 512;  // lanes per workgroup (8 waves)
constexpr int kRowA  = 272;  // pass-A image / exchange: S[row][col], row pitch 272 float2 = 544 dwords = 32 mod 64 banks: rows of different parity never share a bank
constexpr int kRowB  = 513;  // pass-B -> pass-C exchange: S[c][i3], row pitch 513 float2 (1026 dwords = 2 mod 32: 16 c-lanes hit 32 distinct banks)
constexpr int kDPad  = 544; // >= (16 * 15 + 255) * 17 / 16 + 1
constexpr int kSLen  = 32 * kRowA + 16; // 8720 float2 (rows 16..31 are shifted by 16) >= 16 * 513

struct ChainFdArgs {
    const float2* x;      // frames * 8192 samples
    const float2* hist;   // 256 samples preceding x (hist[h] = x[-256 + h])
    const float2* H;      // FFT_8192 of the zero-padded taps
    const float2* twB;    // [16][32]  W_512^{r k}
    const float2* twC;    // [16][512] W_8192^{r i3}
    const float*  taps;   // 256 (zero padded)
    const void*   efrag;  // e on the bf16 matrix pipe: [4 K quarters][2 K-steps][3 tap planes][64 lanes] x 8 bf16, A[j][u] = b_p[256 + j - u]
    const float*  win;    // WIN kernels: window[n] / N (8192 floats; small-FFT mode: the fftSize-point window tiled over the block), else unused
    const float2* twS;    // small-FFT mode: W_fftSize^j
    float*        out;    // frames * 8192 mag2
    long          n_frames;
    float*        pw;     // optional 16 x {sum |x|^2, sum of the outputs' power} + workgroups done + a flag word, over a quarter of every frame's points (dynamic-range guard), else null
    float*        pw_host; // page-locked {in, out}: written with one 8-byte store by the last workgroup to finish (no extra stream operation per launch)
    unsigned      pw_seq;
    unsigned      pw_mask; // 0: every frame is measured
    float         pw_thr; // kModeFir: a frame whose output power is below pw_thr x its input power marks the launch (word 33 of pw, word 3 of pw_host)
    int           no_tier; // the frames this launch marks go straight to the float64 evaluation (counted as such)
    float         pw_c5i; // ... pw_c5i x (peak component of X)^2  -- see kGuardPeakMax
    float         pw_c4;  // the |.|^2 modes (round 6): a frame is marked when  sqrt(sum_k |Y_k|^4)  <  pw_c4 x sum_n |x_n|^2 + ...  -- see kGuardR4Max
    unsigned char* fflags; // optional, one byte per frame (8192-sample block): non-zero = that frame fell below the threshold -- chain_redo_kernel, launched behind this
                           // kernel, evaluates exactly those frames again in the time domain (float64 products): the guard without the host
    // chain_redo_kernel only (0 = the fused kernel's own conventions):
    int            redo_hist_len; // samples `hist` holds in front of x (0: 256)
    int            redo_flags_per_block; // flag bytes per 8192-sample block (0: 1; the fused time-domain kernel judges 4096-sample segments: 2)
    long           redo_n;        // samples of the span (0: n_frames whole blocks); a partial last block reads zeros behind it and writes only what the span holds
    int            redo_min_flag; // chain_redo_kernel takes the blocks whose flag byte is >= this (0: any marked block).  2: chain_td16_kernel has been over the marked frames
                                  // first and left a 2 on the ones whose filter output lies too far below their input for its 22-bit products
    const unsigned short* hfrag;  // chain_td16_kernel: the filter's two-term f16 tap table (fir_f16_make_afrag at KS = 9: fragments + {1 / t, ntaps, sum b^2 / 128, -} + the taps)
    float          td16_thr;      // ... and the fraction of (sum b^2) x the staged input power below which a frame's filter output is left to the float64 evaluation
    unsigned long long* dbg; // GR4_FD_TIMING only
};
// several channels in ONE launch (gr4hip_chain_process_multi, kModeMag2 only).  fold_ch > 1: every workgroup takes frame f of ALL channels in turn (same taps:
// H and the tap fragments are shared) and keeps sum_c |FFT(fir(x_c))|^2 in registers -- the combiner math::Add<float> (blocks/math/.../Math.hpp:73-108, left fold over
// the inputs) as the store epilogue: 8 + 4 / n B per sample instead of 12 + the fold's traffic.  fold_ch == 1: workgroup b belongs to channel b mod n_ch
// (its own taps, history, output and power slots), frames b / n_ch + i gridDim / n_ch: one resident workgroup per CU instead of n persistent kernels that contend for them.
#ifndef GR4_PW_IN_STEP
#define GR4_PW_IN_STEP 1 // every input sample of every frame (a lane's samples t + 512 m are 512-sample stripes of the frame: a subset of m is blind to a burst in the others)
#endif
constexpr int   kLdsEbfBytes = (2 * kSLen + 512 + 256) * 8 + 4 * 2 * 256 * 4 + 6 * 512 * 2; // LDS of the non-windowed filter modes (= lds_ebf of chain_fused_run); the verdict words (kGvBytes) follow it
constexpr int   kGvW = 24, kGvBytes = 2 * kGvW * 4; // verdict words per parity of the frame: eight wave totals of sum |Y|^4, of the input power, eight wave maxima of the input spectrum's peak
constexpr int   kPwFrameSlots = 40, kPwMaxWorkgroups = 2048; // ChainFdArgs::pw: words 0 .. 35 as before, then two words per workgroup for the frames' verdicts
constexpr float kGuardFirFrameThreshold = 0.04f; // the FIR-only fast convolution (fir.hip): its output is y itself, not |Y|^2
constexpr float kTd16Agree = 5.0e-6f;  // chain_td16_kernel stores its spectrum only where it agrees with the fused launch's this closely in every bin (see there)
constexpr float kTd16GuardRatio = 1.0f / 32.0f; // chain_td16_kernel leaves a frame to the float64 evaluation when its filter output carries less than this x (sum b^2) x its input power: 15 dB below what
                                                // white noise would pass, where the 22-bit products' error (~1.3e-7 rms of the products' level, its peaks 4 x that) reaches 6e-6 of |Y|^2 (chain.hip kChainPairGuardRatio)
// Round 6: what a |.|^2 frame is judged on.  The fast convolution's error is K eps-sized relative to the INPUT's level per bin, the parity metric normalises by max(|truth_k|,
// rms_k(truth)) with truth = |Y_k|^2 -- so the error shows as  K sqrt(R4),  R4 = w2 nf mean|x|^2 / rms_k(|Y_k|^2)  (the input's per-bin power over the rms of the OUTPUT mag2
// spectrum; w2 = mean window^2, nf = fftSize), not as a function of the power ratio P_out / P_in: a 1 %-pass-band channel filter over wide-band noise (ratio 0.008, R4 = 6 .. 13:
// error 2.5e-6) is as accurate as a wide one.  Measured per frame with the guard off (tools/dbg/r4_threshold.py: 256 / 129 / 64 taps, cut-offs 0.0025 .. 0.015, rectangular /
// Hann / Blackman-Harris, white noise alone and beside a wide-band neighbour 0 .. 24 dB stronger, steady and in bursts; 90 000 frames; frames the second statistic below marks
// left out): K <= 1.1e-6 everywhere, worst error 3.3e-6 below R4 = 20, 4.8e-6 below 24, 5.3e-6 below 32.  The two statistics' errors add in power, so a frame is marked when
// R4 / kGuardR4Max + T' / kGuardPeakMax > 1: at most max(1.1e-6 sqrt(20), 1.4e-7 sqrt(2000)) = 6.3e-6 on the boundary (white noise carries T' = 13 .. 20 R4: marked from R4 = 16.7).  (Lines -- a rejected tone -- reach
// K = 2.7e-6 through the images of the transform: the second statistic's.)  Until round 6 the power ratio < 0.08 marked them: every frame of every filter that passes less
// than 8 % of white noise (R4 = 2.4 .. 3), where nothing needed fixing.
constexpr float kGuardR4Max = 20.0f;
// The second condition (round 6, found by tools/fuzz_chain.py's wide mode): a float32 transform leaves IMAGES of a strong line -- rounding errors of c eps |X_peak|, c ~ 1.2, at the
// bins N/2 (3N/4, 9N/16 ...: the last radix-16 pass) away from it.  Any float32 FFT has them; the reference's sits behind its filter, ours in front: a line the filter takes down
// by 10 .. 30 dB (in its transition band, or a 2-tap average's single null) still dominates the output's rms, R4 is small, and the image lands in the pass band at full gain next
// to bins of |Y_k|^2 ~ rms: error 2 c eps wg |X_peak| / sqrt(rms).  Measured over the fuzzer's six cases (tools/dbg/case2.py): err <= 1.4e-7 sqrt(T), T = wg^2 |X_peak|^2 /
// rms_k(|Y_k|^2) (wg = mean window), 1.06 .. 1.3e-5 at T = 7e3 .. 3e4.  The kernel takes the peak as the largest |Re| / |Im| of the frame's X (one v_max3 per bin; within
// [1/sqrt2, 1] of |X_peak|) and marks T' = 2 wg^2 peak^2 / rms > kGuardPeakMax, T <= T' <= 2 T: every frame with T >= 2000 (error past 6.3e-6) is marked.  White noise has
// T' = 13 .. 20 R4 (<= 160 where R4 passes), a line IN the pass band ~ sqrt(N) / |H|^2: neither comes near.
constexpr float kGuardPeakMax = 2000.0f;
// fftSize < 8192: the kernel judges 8192-sample BLOCKS (its fast convolution's unit), the metric frames -- a frame shorter than the filter's memory can be 20 dB quieter than its
// block (a 256-point frame behind a 256-tap filter at cut-off 0.0025: the wide fuzzer's 7.7e-6 at block R4 = 13), and a line that sets in late in a block is a smaller peak of the
// block's transform than of its last frames'.  Tighter limits there (R4 at round 6's first, fuzz-validated setting): err / sqrt(T) reaches 2.7e-7 at 256 points.
constexpr float kGuardR4MaxSmall = 8.0f, kGuardPeakMaxSmall = 600.0f;
constexpr float kGuardFrameThreshold = 0.08f; // the guard's output / input power threshold (chain.hip, fir.hip), applied to every frame by itself inside the kernel
constexpr int kMaxMulti = 16;
struct ChainFdMulti {
    int           n_ch, fold_ch;
    const float2* xs[kMaxMulti];
    const float2* hists[kMaxMulti];
    const float2* Hs[kMaxMulti];    // fold_ch == 1 only (else ChainFdArgs::H of the shared taps)
    const void*   efrags[kMaxMulti];
    float*        outs[kMaxMulti];
    float*        pws[kMaxMulti];
    float*        pw_hosts[kMaxMulti];
    unsigned      pw_seqs[kMaxMulti];
    unsigned char* fflags[kMaxMulti]; // fold_ch == 1: every channel's own verdict bytes (one per frame; null: not judged).  The fold's are ChainFdArgs::fflags: a frame is
                                      // marked when ANY channel's share of it fell below the threshold (set-only: the host zeroes the bytes in front of the launch)
};

// pass-A layout: rows 0..15 at r * kRowA, rows 16..31 shifted by 16 float2 (32 banks).  A ds_read/write_b64 is served in two groups of 32
// lanes over 64 four-byte banks; every access pattern of the kernel puts the two 16-lane halves of a group 32 banks apart:
//   pass-A reads   lane pair = rows 2m, 2m+1 of one column (different parity: one pitch = 32 banks)
//   pass-A writes  lane pair = rows k1, k1+16 (the shift)
//   pass-B gathers / scatters  the 32-lane group holds rows k and k+16 (gather: the shift; scatter S2[c][32q + k]: 2 * 16 = 32 banks)
__device__ __forceinline__ int addrA(int row, int col) { return row * kRowA + col + ((row & 16) ? 16 : 0); }

// pass B (p = 32, radix 16): butterfly (c, k) gathers S1[k][c + 16 r], twiddles by W_512^{r k}, scatters to S2[c][32 q + k]
__device__ __forceinline__ void passB_compute_store(float2* S, float2 (&w)[16], const float2 (&tw)[16], int c, int k) {
#pragma unroll
    for (int r = 1; r < 16; ++r) w[r] = cmul(w[r], tw[r]);
    fft16<1>(w);
#pragma unroll
    for (int q = 0; q < 16; ++q) S[c * kRowB + 32 * q + k] = w[perm16(q)];
}
// same, twiddles from an LDS table [16][32] (the windowed kernel has no registers left for a resident set)
__device__ __forceinline__ void passB_table_store(float2* S, float2 (&w)[16], const float2* twl, int c, int k) {
#pragma unroll
    for (int r = 1; r < 16; ++r) w[r] = cmul(w[r], twl[r * 32 + k]);
    fft16<1>(w);
#pragma unroll
    for (int q = 0; q < 16; ++q) S[c * kRowB + 32 * q + k] = w[perm16(q)];
}
// pass C (p = 512, radix 16): butterfly i3 gathers S2[r][i3], twiddles by W_8192^{r i3}; X[i3 + 512 q] ends at slot perm16(q)
__device__ __forceinline__ void passC(const float2* S, float2 (&g)[16], const float2 (&tw)[16], int i3) {
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = S[r * kRowB + i3];
#pragma unroll
    for (int r = 1; r < 16; ++r) g[r] = cmul(g[r], tw[r]);
    fft16<1>(g);
}

// phase fence: keeps hipcc's scheduler from overlapping independent phases (which costs more registers than the 128 a lane has
// at 4 waves per SIMD and ends in scratch spills)
#define GR4_PHASE_FENCE()                      \
    do {                                       \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

// developer instrumentation (-DGR4_FD_TIMING): wave 0 of every workgroup stamps s_memtime at phase boundaries into a.dbg
#ifdef GR4_FD_TIMING
#define GR4_STAMP(i)                                                                                  \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
        if ((threadIdx.x & 63) == 0 && a.dbg) a.dbg[(((f / gridDim.x) * gridDim.x + blockIdx.x) * 8 + (threadIdx.x >> 6)) * 16 + (i)] = t_; \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
#else
#define GR4_STAMP(i) do { } while (0)
#endif
#ifdef GR4_FD_TIMING
#define GR4_PIN(v) pin16(v)
#else
#define GR4_PIN(v) do { } while (0)
#endif

// DMA pieces (of 8 per wave) issued at drain point g: [kDmaAt[g], kDmaAt[g + 1]) -- front-loaded, the last piece leaves ~2/3 of a
// frame time before it is needed
#ifndef GR4_DMA_SCHED
#define GR4_DMA_SCHED 0
#endif
#if GR4_DMA_SCHED == 0
constexpr int kDmaAt[9] = {0, 1, 2, 3, 4, 5, 6, 7, 8};
#elif GR4_DMA_SCHED == 1
constexpr int kDmaAt[9] = {0, 0, 1, 2, 3, 5, 6, 7, 8};
#elif GR4_DMA_SCHED == 2
constexpr int kDmaAt[9] = {0, 1, 1, 3, 4, 5, 6, 7, 8};
#else
constexpr int kDmaAt[9] = {0, 0, 0, 2, 4, 5, 6, 7, 8};
#endif

typedef __attribute__((address_space(3))) void*       lds_ptr_t;
typedef __attribute__((address_space(1))) const void* gbl_ptr_t;

// force 16 complex values into VGPRs at this point (any spill reload happens HERE, i.e. before an LDS-DMA is put in flight:
// a scratch reload behind the DMA makes hipcc wait vmcnt(0) and serialises the prefetch with the tail of the frame)
__device__ __forceinline__ void pin16(float2 (&v)[16]) {
    asm volatile("" : "+v"(v[0].x), "+v"(v[0].y), "+v"(v[1].x), "+v"(v[1].y), "+v"(v[2].x), "+v"(v[2].y), "+v"(v[3].x), "+v"(v[3].y),
                      "+v"(v[4].x), "+v"(v[4].y), "+v"(v[5].x), "+v"(v[5].y), "+v"(v[6].x), "+v"(v[6].y), "+v"(v[7].x), "+v"(v[7].y));
    asm volatile("" : "+v"(v[8].x), "+v"(v[8].y), "+v"(v[9].x), "+v"(v[9].y), "+v"(v[10].x), "+v"(v[10].y), "+v"(v[11].x), "+v"(v[11].y),
                      "+v"(v[12].x), "+v"(v[12].y), "+v"(v[13].x), "+v"(v[13].y), "+v"(v[14].x), "+v"(v[14].y), "+v"(v[15].x), "+v"(v[15].y));
}

// One LDS-DMA piece: 64 lanes x 16 B = 1 KiB from per-lane global addresses to LDS [lds_byte_addr, +1 KiB) (M0 = wave-uniform LDS
// address).  Issued from inline asm ON PURPOSE: hipcc's wait insertion does not see it, so it neither drains the copy in front of
// the next ds_read (it cannot prove that the two frame buffers do not alias) nor in front of a barrier.  The copy is ordered by
// the explicit s_waitcnt vmcnt(0) + s_barrier at the top of the frame loop (GR4_FULL_BARRIER), nothing else.
__device__ __forceinline__ void dma_1k(const void* gsrc_lane, unsigned lds_byte_addr) {
    unsigned keep;
    lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr); // provably wave-uniform for the "s" constraint
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" GR4_DMA_MODS "\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc_lane), "s"(lds_byte_addr)
                 : "memory");
}
__device__ __forceinline__ unsigned lds_addr(const void* p) { return (unsigned)(uintptr_t)(lds_ptr_t)p; }

// asynchronous HBM -> LDS copy of one frame (64 KB) into the pass-A exchange layout: 64 one-KiB pieces, 8 per wave; no VGPR
// staging, the data is in flight while the workgroup transforms the previous frame.
template <int I0 = 0, int I1 = 8>
__device__ __forceinline__ void dma_frame(const float2* __restrict__ xf, float2* S, int wave, int lane) {
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_addr(S));
#pragma unroll
    for (int i = I0; i < I1; ++i) {
        const int row = 4 * wave + (i >> 1), half = i & 1;
        dma_1k(xf + row * 256 + 128 * half + 2 * lane, base + (unsigned)addrA(row, 128 * half) * 8u);
    }
}

// the 256 samples preceding frame f (previous frame's tail, or the carried history for the first frame of a call): 2 KiB, waves 0/1
__device__ __forceinline__ void dma_tail(const float2* __restrict__ src, float2* Tl, int wave, int lane) {
    const unsigned base = __builtin_amdgcn_readfirstlane(lds_addr(Tl));
    if (wave < 2) dma_1k(src + 128 * wave + 2 * lane, base + 1024u * wave);
}

// workgroup barriers that do NOT drain the vector-memory counter: an LDS-DMA for the next frame stays in flight across them
// (__syncthreads() would emit s_waitcnt vmcnt(0) while a DMA is pending and serialise the prefetch with this frame's work)
#if defined(GR4_T_NOBAR) // developer timing build (results are wrong): the in-frame barriers without the s_barrier = the bound of hiding every in-frame barrier wait behind other work
#define GR4_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#else
#define GR4_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#endif
#define GR4_FULL_BARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")

// pass A on an image already in LDS, in place (the second and third transforms of the windowed kernel)
__device__ __forceinline__ void passA_inplace(float2* S, const float2 (&twA)[16], int par, int n0, float sgn) {
    float2 v[16];
#pragma unroll
    for (int m = 0; m < 16; ++m) v[m] = S[addrA(2 * m + par, n0)];
    fft16<1>(v);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
        const float2 u = k1 == 0 ? v[perm16(k1)] : cmul(v[perm16(k1)], twA[k1]); // twA: W_32^k1 on the odd lane, 1 on the even lane
        const float2 q = make_float2(lane_xor1(u.x), lane_xor1(u.y));
        S[addrA(k1 + 16 * par, n0)] = make_float2(fmaf(sgn, u.x, q.x), fmaf(sgn, u.y, q.y));
    }
}

// One persistent workgroup of 512 lanes (8 waves, 2 per SIMD) per CU; frame f = blockIdx.x, blockIdx.x + gridDim.x, ...
// Every lane owns 16 points.  Two frame buffers alternate: while frame f is transformed in one, the LDS-DMA of frame
// f + gridDim.x fills the other during the WHOLE frame time (smooth HBM demand instead of chip-wide bursts).  Nothing in the
// loop issues an ordinary vector load behind the DMA (loads return in order, so waiting for one would wait for the DMA):
// H[k], the twiddle bases and the FIR taps live in registers / SGPRs for the lifetime of the kernel.
//
// WIN = true (any window other than None / Rectangular; the reference FFT block's default is Hann): the window multiplies the FILTERED
// frame, so y_f has to exist in the time domain: X = FFT(x_f), y_f = IFFT(H X) + e (inverse through conjugation, e added where it
// lives), then FFT(w y_f): three full transforms instead of 1.65, still without touching HBM in between.
//
// MODE 2 stops after the inverse transform and writes y_f itself: fir_filter<complex<float>> as a fast convolution (2 transforms
// per 8192 samples instead of 1024 flop per sample), used by gr4hip_fir_process for long complex inputs.
//
// MODE 3 (kModeWinSmall, LOG2NF = 8..12): the FFT block runs at fftSize = 2^LOG2NF < 8192 (its default is 1024).  The FIR part is unchanged --
// 8192-sample blocks, y recovered in the time domain as in MODE 1 -- and the third transform becomes 8192 / fftSize independent windowed
// fftSize-point transforms of the block (the compile-time 16 x 16 x R3 plan of the FFT block kernels, fft_radix.hpp, on the LDS image).
// MODE 4 / 5 (kModeFftMag2 / kModeFftWinMag2): no filter at all -- the FFT block's |X|^2 output at fftSize 8192 (gr4hip_fft_mag2) on this kernel's frame pipeline:
// the next frame streams in by LDS-DMA while this one is transformed, which the load -> transform -> store body of fft_fast_kernel cannot do.
// MODE 6 / 7 (kModeFftSpec / kModeFftWinSpec): the same plain transform, the complex spectrum itself as output (gr4hip_fft_spectrum).
enum { kModeMag2 = 0, kModeWinMag2 = 1, kModeFir = 2, kModeWinSmall = 3, kModeFftMag2 = 4, kModeFftWinMag2 = 5, kModeFftSpec = 6, kModeFftWinSpec = 7 };
// sum of v over the wave on the DPP network (row shifts, then the two row broadcasts): the total lands in lane 63
__device__ __forceinline__ float wave_total_lane63(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x111, 0xf, 0xf, true)); // row_shr:1
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x112, 0xf, 0xf, true)); // row_shr:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x114, 0xf, 0xf, true)); // row_shr:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x118, 0xf, 0xf, true)); // row_shr:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x142, 0xa, 0xf, true)); // row_bcast:15 into rows 1, 3
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x143, 0xc, 0xf, true)); // row_bcast:31 into rows 2, 3
    return v;
}

// maximum of v (>= 0) over the wave, same network: lands in lane 63.  On the bit patterns (floats >= 0 order as unsigned integers): fmaxf would canonicalise every operand
// (a v_max_f32 x, x each) and keep the DPP move apart from the maximum
__device__ __forceinline__ float wave_max_lane63(float vf) {
    unsigned v = __float_as_uint(vf);
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x111, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x112, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true));
    v = max(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true));
    return __uint_as_float(v);
}
// max(acc, |x|, |y|) in one instruction (fmaxf(fabsf()) costs three: the compiler canonicalises each operand first)
__device__ __forceinline__ float max3_abs(float acc, float x, float y) {
    asm("v_max3_f32 %0, %0, |%1|, |%2|" : "+v"(acc) : "v"(x), "v"(y));
    return acc;
}
// the largest of eight floats >= 0 (two float4 of LDS), on their bit patterns
__device__ __forceinline__ float max8_nonneg(const float4& a, const float4& b) {
    const unsigned m0 = max(max(__float_as_uint(a.x), __float_as_uint(a.y)), max(__float_as_uint(a.z), __float_as_uint(a.w)));
    const unsigned m1 = max(max(__float_as_uint(b.x), __float_as_uint(b.y)), max(__float_as_uint(b.z), __float_as_uint(b.w)));
    return __uint_as_float(max(m0, m1));
}

template <int MODE, int LOG2NF, bool MULTI>
__device__ __forceinline__ void chain_fd_body(ChainFdArgs& a, const ChainFdMulti* mc) {
    constexpr bool WIN   = MODE == kModeWinMag2 || MODE == kModeWinSmall; // y_f is needed in the time domain and multiplied by a.win
    constexpr bool SMALL = MODE == kModeWinSmall;
    constexpr bool FFTONLY = MODE == kModeFftMag2 || MODE == kModeFftWinMag2 || MODE == kModeFftSpec || MODE == kModeFftWinSpec; // plain (windowed) transform: no taps, no history, no correction
    constexpr bool FFTWIN  = MODE == kModeFftWinMag2 || MODE == kModeFftWinSpec;
    constexpr bool SPEC    = MODE == kModeFftSpec || MODE == kModeFftWinSpec;
    constexpr bool FIR   = MODE == kModeFir || SPEC;                           // complex output (y_f, or the spectrum): 8 bytes per sample
    constexpr bool DEFER = true;                                             // the previous frame's results leave during this frame's phases
    static_assert(!MULTI || MODE == kModeMag2, "several channels per launch: headline mode only");
    // frames of this workgroup: fstart, fstart + fstride, ...; gslot / gcount: its place among the workgroups that report to a.pw
    long     fstart = blockIdx.x, fstride = gridDim.x;
    unsigned gslot = blockIdx.x, gcount = gridDim.x;
    int      C = 1; // channels folded by this workgroup (MULTI with shared taps), else 1
    if constexpr (MULTI) {
        if (mc->fold_ch > 1) {
            C = mc->fold_ch;
        } else { // this workgroup belongs to ONE channel: its pointers take the place of the single-channel arguments
            const int c = (int)(blockIdx.x % (unsigned)mc->n_ch);
            a.x = mc->xs[c]; a.hist = mc->hists[c]; a.H = mc->Hs[c]; a.efrag = mc->efrags[c]; a.out = mc->outs[c]; a.pw = mc->pws[c]; a.pw_host = mc->pw_hosts[c]; a.pw_seq = mc->pw_seqs[c];
            a.fflags = mc->fflags[c];
            fstart = blockIdx.x / (unsigned)mc->n_ch;
            fstride = gridDim.x / (unsigned)mc->n_ch;
            gslot = (unsigned)fstart; gcount = (unsigned)fstride;
        }
    }
    auto xof    = [&](int c) -> const float2* { if constexpr (MULTI) { if (C > 1) return mc->xs[c]; } return a.x; };
    auto histof = [&](int c) -> const float2* { if constexpr (MULTI) { if (C > 1) return mc->hists[c]; } return a.hist; };
    extern __shared__ __attribute__((aligned(16))) float2 smem[]; // the ONLY LDS object (a second one would make hipcc drain the DMA early)
    float2* B0 = smem;                               // kSLen: frame image / exchange buffer (even frames of this workgroup)
    float2* B1 = smem + kSLen;                       // kSLen: (odd frames)
    float2* T0 = B1 + kSLen;                         // 256: tail of the stream before the frame in B0 (x[fN - 256 + j])
    float2* T1 = T0 + 256;                           // 256: same for B1
    float2* el = T1 + 256;                           // 256: e[n]
    float*  P  = reinterpret_cast<float*>(el + 256); // [4 K quarters][re, im][256]: partial e; WIN: followed by the pass-B twiddle table [16][32]
    if constexpr (!WIN && !FFTONLY) { // the frames' verdict words (dynamic-range guard) start at zero; the first frame's top barrier comes before anyone reads them
        if (threadIdx.x < 2 * kGvW) reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + kLdsEbfBytes)[threadIdx.x] = 0.f;
    }
    constexpr bool EBFW = GR4_E_BF16 && GR4_E_BF16_WIN && WIN;                  // windowed modes: the table keeps rows 1 .. 15 only (row 0 is never read), which is the 256 bytes the bf16 planes need
    float*  Dre = P + 4 * 2 * 256 + (WIN ? (EBFW ? 960 : 1024) : 0);                  // kDPad: Dz[s] = d[s - 1] (1 <= s <= 255), zero elsewhere (s <= 511); planar, one pad
    float*  Dim = Dre + kDPad;                       //        float per 16 samples so that the MFMA B-operand reads are conflict-free
    float*  hl  = Dim + kDPad;                       // 272: taps, zero from 256 on
    // EBF: e on the bf16 matrix pipe with three-term splits (fir_bf16.hip) -- the f32 MFMA shares the VALU's issue slot, and this
    // kernel is VALU-bound.  The same LDS bytes then hold six bf16 planes of Dz (re h, m, l, im h, m, l; 512 elements each, zero outside 1 .. 255) instead of Dre / Dim / hl.
    constexpr bool EBF = GR4_E_BF16 && (!WIN || EBFW) && !FFTONLY;
    unsigned short* Dp = reinterpret_cast<unsigned short*>(Dre);

    const int t0   = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(t0 >> 6), lane0 = t0 & 63;

    if constexpr (EBF) {
        for (int i = t0; i < 6 * 512 / 2; i += kT) reinterpret_cast<unsigned*>(Dp)[i] = 0u;
    } else if constexpr (!FFTONLY) {
        for (int i = t0; i < 2 * kDPad; i += kT) Dre[i] = 0.f; // Dre and Dim are adjacent
        if (t0 < 272) hl[t0] = t0 < 256 ? a.taps[t0] : 0.f;
    }
    using e_u32x4 = __attribute__((ext_vector_type(4))) unsigned;
    e_u32x4 ea[2][3]; // EBF: this wave's tap fragments (K quarter wave & 3), kernel-lifetime registers
    if constexpr (EBF) {
        const e_u32x4* ef = static_cast<const e_u32x4*>(a.efrag) + (size_t)(wave & 3) * 2 * 3 * 64 + lane0;
#pragma unroll
        for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
            for (int p = 0; p < 3; ++p) ea[s_][p] = ef[(s_ * 3 + p) * 64];
    }

    // pass A roles: column n0, parity par (even / odd rows of the column); the pair (2 n0, 2 n0 + 1) are neighbouring lanes
    // pass B roles: c = lane & 15, k = 2 wave + {0, 16, 1, 17}[lane >> 4]  (k and k + 16 share a 32-lane group: disjoint banks)
    const int kq0 = lane0 >> 4;
    const int kb0 = 2 * wave + (kq0 >> 1) + 16 * (kq0 & 1);
    // ---- kernel-lifetime registers: H[t + 512 q], twiddle bases W_512^{kb}, W_512^{2 kb}, W_8192^{t}, W_8192^{2t}
    float2 Hr[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) Hr[q] = FFTONLY ? make_float2(1.f, 0.f) : a.H[t0 + 512 * q];
    float2 twBr[16], twCr[16]; // exact table values, resident for the kernel lifetime (no per-frame twiddle generation)
#pragma unroll
    for (int r = 1; r < 16; ++r) {
        twBr[r] = WIN ? make_float2(0.f, 0.f) : a.twB[r * 32 + kb0]; // WIN: pass-B twiddles come from the LDS table twBl instead
        twCr[r] = a.twC[r * 512 + t0];
    }
    // pass-A pair twiddle as a per-lane value (W_32^k1 on the odd lane of a pair, 1 on the even one): one multiply for both lanes instead
    // of multiply + two selects
    float2 twA[16];
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) twA[k1] = (t0 & 1) ? w32(k1) : make_float2(1.f, 0.f);
    // WIN: the correction FIR splits K over 4 wave pairs (one of each pair takes the real tile, the other the imaginary one), so only
    // half of P is used; the other half holds the pass-B twiddle table
    float2* twBl = reinterpret_cast<float2*>(P + 4 * 2 * 256) - (EBFW ? 32 : 0);
    if constexpr (WIN) { if (!EBFW || t0 >= 32) twBl[t0] = a.twB[t0]; } // [16][32] = 512 entries (EBFW: rows 1 .. 15)

    float wA[16]; // kModeFftWinMag2: window of the samples pass A reads, row 2 m + par of column n0 = sample (2 m + par) 256 + n0
    if constexpr (FFTWIN) {
#pragma unroll
        for (int m = 0; m < 16; ++m) wA[m] = a.win[(2 * m + (t0 & 1)) * 256 + (t0 >> 1)];
    }
    float wr[16]; // WIN: window[t + 512 q] / N
    if constexpr (WIN) {
#pragma unroll
        for (int q = 0; q < 16; ++q) wr[q] = a.win[t0 + 512 * q];
    }
    // SMALL: twiddle bases of the fftSize-point plan for this lane (lane tt of its frame's fftSize / 16 lanes)
    constexpr int NF = 1 << (SMALL ? LOG2NF : 8), TF = NF / 16, NPF = NF + NF / 32;
    float2        sw2a = make_float2(1.f, 0.f), sw2b = sw2a, sw3 = sw2a, sw3sq = sw2a;
    if constexpr (SMALL) {
        const int tt = t0 % TF;
        sw2a  = a.twS[(tt & 15) * (NF / 256)];
        sw2b  = a.twS[2 * (tt & 15) * (NF / 256)];
        sw3   = a.twS[tt & 255];
        sw3sq = a.twS[(2 * (tt & 255)) & (NF - 1)];
    }

    // |Y|^2 of the previous frame waits in registers and leaves in four groups of four stores spread over this frame's phases;
    // likewise the eight DMA pieces per wave of the next frame.  A wave that issues its 16 stores (or 8 DMAs) back to back sits
    // in VMEM issue for ~4000 cycles behind the other waves' requests and every barrier inherits the skew.
    float pend[16], pendi[16]; // (pendi: imaginary parts, kModeFir only -- y_f is complex)
    float pw_in = 0.f, pw_out = 0.f; // dynamic-range guard: sampled input and output power of this workgroup's frames
    // ... and every frame's own verdict: the sum over the WORKGROUP of out - thr * in (a wave alone will not do: its bins are 64-bin windows 2048 apart, and a narrow
    // pass band lands in the windows of two or three waves).  Each wave adds its total into one of two words of global scratch (L2 atomics, no LDS: the windowed
    // kernels have none to spare, and no barrier of their own: the frame's top barrier orders them), thread 0 collects the sum of the frame before two iterations later.
    float  pw_dprev = 0.f;  // this lane's out - thr * in of the frame before (kModeFir) / its sum of |Y_k|^4 (the |.|^2 modes)
    [[maybe_unused]] float pw_iprev = 0.f; // the |.|^2 modes: this lane's share of the frame's input power
    [[maybe_unused]] float pw_kprev = 0.f; // the |.|^2 modes: the largest |Re| / |Im| among this lane's bins of the frame's input spectrum
    [[maybe_unused]] float pw_pend_i = 0.f;
    float  pw_pend = 0.f;   // thread 0: the workgroup sum requested an iteration ago
    float  pw_dmin = 0.f;   // thread 0: the smallest workgroup sum so far (negative: a frame fell below the threshold by itself)
    unsigned pw_nmark = 0;  // thread 0: how many of this workgroup's frames (MULTI: work items) were marked
    float* pw_slot = a.pw != nullptr ? a.pw + kPwFrameSlots + 6 * blockIdx.x : nullptr;
    constexpr bool Q4 = MODE != kModeFir; // judged on the fourth-moment statistic
    // the frame's verdict from its two workgroup sums: negative = marked (a sum that left float32's range marks too)
    // (sa: sum of |Y_k|^4, sb: input power, sk: peak component of the input spectrum -- the two conditions of kGuardR4Max / kGuardPeakMax)
    // R4 / kGuardR4Max + T' / kGuardPeakMax > 1 (the two errors add in power):  sqrt(S4) < pw_c4 x S_in + pw_c5i x peak^2  (an overflow on the right marks; so does a sum on the left that left float32's range)
    const auto verdict = [&](float sa, float sb, float sk) { return Q4 ? ((sa < 3.0e38f) ? __builtin_sqrtf(sa) - fmaf(sb, a.pw_c4, (sk * sk) * a.pw_c5i) : -1.f) : sa; };
    int   iter = 0;
    [[maybe_unused]] int fiter = 0; // frames this workgroup has finished (MULTI: an item is one channel of a frame)
#pragma unroll
    for (int q = 0; q < 16; ++q) pend[q] = pendi[q] = 0.f;
    long fprev = -1, fh1 = -1, fh2 = -1;
    long f   = fstart;
    int  cur = 0, ch = 0; // ch: channel of the current work item (MULTI with shared taps: frame f of channels 0 .. C - 1 in turn)
#if defined(GR4_PRIO_YOUNG)
    if (wave >= 4) __builtin_amdgcn_s_setprio(1); // the second-dispatched half loses every age arbitration on its SIMD otherwise
#endif
    if (f < a.n_frames) {
        if constexpr (!FFTONLY) dma_tail(f > 0 ? xof(0) + f * kN - 256 : histof(0), T0, wave, lane0);
        dma_frame(xof(0) + f * kN, B0, wave, lane0);
    }
    for (; f < a.n_frames; cur ^= 1, ++iter) {
        // EVERY frame: an eighth of its input samples, all of its output bins (about 35 VALU instructions per lane and frame): a frame the guard does not look at is a frame it cannot vouch for.
        // (pw_mask is 0; as a run-time value it keeps the branch a branch -- with a loop-invariant condition the compiler builds two loops and the measured one spills)
        const bool measure = !FFTONLY && a.pw != nullptr && ((unsigned)iter & a.pw_mask) == 0u;
        [[maybe_unused]] float fr_in = 0.f; // this work item's sampled input power
        // per-iteration opaque copy of the lane id: lane-dependent LDS / buffer offsets are recomputed here (a few VALU ops)
        // instead of being hoisted out of the loop as dozens of loop-invariant VGPRs
        int tl = threadIdx.x;
        asm volatile("" : "+v"(tl));
        const int   lane = tl & 63, t = tl;
        const int   n0 = tl >> 1, par = tl & 1;
        const float sgn = par ? -1.f : 1.f;
        const int   cb = lane & 15, kq = lane >> 4;
        const int   kb = 2 * wave + (kq >> 1) + 16 * (kq & 1);
        float2* S  = cur ? B1 : B0;
        float2* Sn = cur ? B0 : B1;
        [[maybe_unused]] float m2sum = 0.f, m4sum = 0.f;
        const float2* Tc = cur ? T1 : T0;
        GR4_STAMP(0);
        GR4_FULL_BARRIER(); // T: this frame's image has landed (vmcnt(0) + barrier); everything of the previous frame is dead
        GR4_STAMP(1);
        if constexpr (!FFTONLY) {
            if (a.pw != nullptr) {
                const float wt = wave_total_lane63(pw_dprev); // the frame before this one
                [[maybe_unused]] float wti = 0.f, wtk = 0.f;
                if constexpr (Q4) { wti = wave_total_lane63(pw_iprev); wtk = wave_max_lane63(pw_kprev); }
                if constexpr (!WIN) { // 192 bytes of LDS behind the image: wave totals of frame i - 1 in, the sums of frame i - 2 out (wave 0); the frame's own barriers order them
                    float* Gv = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + kLdsEbfBytes);
                    if ((threadIdx.x & 63) == 63) { Gv[kGvW * (iter & 1) + (threadIdx.x >> 6)] = wt; if constexpr (Q4) { Gv[kGvW * (iter & 1) + 8 + (threadIdx.x >> 6)] = wti; Gv[kGvW * (iter & 1) + 16 + (threadIdx.x >> 6)] = wtk; } }
                    if (threadIdx.x < 64) {
                        const float4 g0 = *reinterpret_cast<const float4*>(Gv + kGvW * ((iter + 1) & 1)), g1 = *reinterpret_cast<const float4*>(Gv + kGvW * ((iter + 1) & 1) + 4);
                        float fsum = ((g0.x + g0.y) + (g0.z + g0.w)) + ((g1.x + g1.y) + (g1.z + g1.w)); // the frame two iterations back
                        if constexpr (Q4) {
                            const float4 h0 = *reinterpret_cast<const float4*>(Gv + kGvW * ((iter + 1) & 1) + 8), h1 = *reinterpret_cast<const float4*>(Gv + kGvW * ((iter + 1) & 1) + 12);
                            const float4 k0 = *reinterpret_cast<const float4*>(Gv + kGvW * ((iter + 1) & 1) + 16), k1 = *reinterpret_cast<const float4*>(Gv + kGvW * ((iter + 1) & 1) + 20);
                            fsum = verdict(fsum, ((h0.x + h0.y) + (h0.z + h0.w)) + ((h1.x + h1.y) + (h1.z + h1.w)), max8_nonneg(k0, k1));
                        }
                        pw_dmin = fminf(pw_dmin, fsum);
                        pw_nmark += (unsigned)(fsum < 0.f);
                        if (a.fflags != nullptr && threadIdx.x == 0 && fh2 >= 0) {
                            if constexpr (MULTI) { if (fsum < 0.f) a.fflags[fh2] = 1; } // (several work items per frame: set-only)
                            else a.fflags[fh2] = fsum < 0.f ? 1 : 0;
                        }
                    }
                } else { // the windowed kernels fill their 160 KiB to the byte: two words of global scratch per workgroup, L2 atomics
                    pw_dmin = fminf(pw_dmin, pw_pend);
                    pw_nmark += (unsigned)(pw_pend < 0.f);
                    if ((threadIdx.x & 63) == 63) { atomicAdd(pw_slot + (iter & 1), wt); if constexpr (Q4) { atomicAdd(pw_slot + 2 + (iter & 1), wti); atomicMax(reinterpret_cast<unsigned*>(pw_slot) + 4 + (iter & 1), __float_as_uint(wtk)); } } // (floats >= 0 order as their bit patterns)
                    if (threadIdx.x == 0) {
                        pw_pend = atomicExch(pw_slot + ((iter + 1) & 1), 0.f); // the frame before that: every wave's share arrived before the barrier above
                        if constexpr (Q4) { const float si2 = atomicExch(pw_slot + 2 + ((iter + 1) & 1), 0.f); pw_pend = verdict(pw_pend, si2, __uint_as_float(atomicExch(reinterpret_cast<unsigned*>(pw_slot) + 4 + ((iter + 1) & 1), 0u))); } // (a slot nothing was filed in holds zeros: verdict 0, not marked)
                        if (a.fflags != nullptr && fh2 >= 0) a.fflags[fh2] = pw_pend < 0.f ? 1 : 0; // (the windowed modes are single-channel)
                    }
                }
            }
        }
        // stream the next frame (and the 256 samples before it) into the other buffers: in flight until the next barrier T.
        // Unconditional (the last iteration re-reads its own frame): no divergent paths around the DMA for hipcc's wait insertion
#if defined(GR4_T_L2ONLY) // developer timing build (results are wrong): every workgroup re-reads and re-writes ITS FIRST frame -- the same instructions with the traffic held in L2
        const long    fn  = blockIdx.x;
        const int     chn = 0;
        const float2* xn  = a.x;
        const rsrc_t  rq  = make_rsrc(a.out + (long)blockIdx.x * kN * (FIR ? 2 : 1), fprev < 0 ? 0u : (unsigned)(kN * sizeof(float) * (FIR ? 2 : 1)));
#else
        long fn  = (f + fstride < a.n_frames) ? f + fstride : f;
        int  chn = 0;
        if constexpr (MULTI) { // next work item: the same frame of the next channel, then the workgroup's next frame of channel 0
            chn = ch + 1;
            fn  = f;
            if (chn >= C) { chn = 0; fn = f + fstride; }
            if (fn >= a.n_frames) { fn = f; chn = ch; }
        }
        const float2* xn = xof(chn);
        const rsrc_t rq = make_rsrc(a.out + (fprev < 0 ? 0 : fprev) * kN * (FIR ? 2 : 1), fprev < 0 ? 0u : (unsigned)(kN * sizeof(float) * (FIR ? 2 : 1))); // first iteration: nothing pending, stores fall out of range
#endif
#define GR4_DRAIN(g)                                                                                       \
    do {                                                                                                   \
        if constexpr (DEFER && FIR) { _Pragma("unroll") for (int q = 2 * (g); q < 2 * (g) + 2; ++q) buf_store_f2(rq, make_float2(pend[q], pendi[q]), t * 8, q * 4096); } \
        else if constexpr (DEFER && SMALL) { _Pragma("unroll") for (int q = 2 * (g); q < 2 * (g) + 2; ++q) buf_store_f(rq, pend[q], ((t / TF) * NF + t % TF) * 4, q * TF * 4); } \
        else if constexpr (DEFER) { _Pragma("unroll") for (int q = 2 * (g); q < 2 * (g) + 2; ++q) buf_store_f(rq, pend[q], t * 4, q * 2048); } \
        dma_frame<kDmaAt[g], kDmaAt[(g) + 1]>(xn + fn * kN, Sn, wave, lane);                               \
    } while (0)
        if constexpr (!FFTONLY) dma_tail(fn > 0 ? xn + fn * kN - 256 : histof(chn), cur ? T0 : T1, wave, lane);
        GR4_DRAIN(0);

        // ------------------------------------------------------------------ pass A: 32-point DFT down the 256 columns, in place
        // (a lane pair reads all 32 rows of its column before it writes them back: no barrier needed)
        {
            float2 v[16];
#pragma unroll
            for (int m = 0; m < 16; ++m) v[m] = S[addrA(2 * m + par, n0)];
            if constexpr (FFTWIN) {
#pragma unroll
                for (int m = 0; m < 16; ++m) v[m] = make_float2(v[m].x * wA[m], v[m].y * wA[m]);
            }
            if constexpr (!FFTONLY) {
                if (measure) { // every input sample of the frame (GR4_PW_IN_STEP 1), re and im in one v_pk_fma_f32: the lane's samples sit in the register pairs ds_read_b64 filled
                    using f32x2 = __attribute__((ext_vector_type(2))) float;
                    f32x2 s2 = {0.f, 0.f};
#pragma unroll
                    for (int m = 0; m < 16; m += GR4_PW_IN_STEP) { const f32x2 vm = {v[m].x, v[m].y}; s2 = __builtin_elementwise_fma(vm, vm, s2); }
                    const float s = s2.x + s2.y;
                    fr_in = (float)GR4_PW_IN_STEP * s;
                    pw_in += fr_in;
                }
                if (par && n0 > 0) { // v[15] is x_f[N - 256 + n0]:  Dz[n0] = d[n0 - 1]
                    const float2 dd = csub(Tc[n0], v[15]);
                    if constexpr (EBF) {
                        const __bf16 rh = (__bf16)dd.x, ih = (__bf16)dd.y;
                        const float  r1 = dd.x - (float)rh, i1 = dd.y - (float)ih;
                        const __bf16 rm = (__bf16)r1, im = (__bf16)i1;
                        const __bf16 rl = (__bf16)(r1 - (float)rm), il = (__bf16)(i1 - (float)im);
                        Dp[n0]           = __builtin_bit_cast(unsigned short, rh);
                        Dp[512 + n0]     = __builtin_bit_cast(unsigned short, rm);
                        Dp[2 * 512 + n0] = __builtin_bit_cast(unsigned short, rl);
                        Dp[3 * 512 + n0] = __builtin_bit_cast(unsigned short, ih);
                        Dp[4 * 512 + n0] = __builtin_bit_cast(unsigned short, im);
                        Dp[5 * 512 + n0] = __builtin_bit_cast(unsigned short, il);
                    } else {
                        Dre[n0 + (n0 >> 4)] = dd.x;
                        Dim[n0 + (n0 >> 4)] = dd.y;
                    }
                }
            }
            fft16<1>(v); // even lanes: E16[k1], odd lanes: O16[k1], at slot perm16(k1)
            GR4_DRAIN(1);
#pragma unroll
            for (int k1 = 0; k1 < 16; ++k1) {
                // X[k1] = E + W O (even lane), X[k1 + 16] = E - W O (odd lane), W = W_32^k1
                const float2 u = k1 == 0 ? v[perm16(k1)] : cmul(v[perm16(k1)], twA[k1]); // twA: W_32^k1 on the odd lane, 1 on the even lane
                const float2 q = make_float2(lane_xor1(u.x), lane_xor1(u.y));
                S[addrA(k1 + 16 * par, n0)] = make_float2(fmaf(sgn, u.x, q.x), fmaf(sgn, u.y, q.y));
            }
        }
        GR4_STAMP(2);
        GR4_LDS_BARRIER(); // #1
        GR4_STAMP(3);
        GR4_DRAIN(2);

        GR4_DRAIN(3);
        // ------------------------------------------------------------------ X pass B (p = 32, radix 16)
        float2 w[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) w[r] = S[addrA(kb, cb + 16 * r)];
        GR4_PIN(w);
        GR4_STAMP(4);
        GR4_LDS_BARRIER(); // #2
        GR4_STAMP(5);
        GR4_DRAIN(4);
        // ------------------------------------------------------------------ e[n] = sum_j b[j] Dz[256 + n - j] on the MFMA units
        // Block-Toeplitz form with n = 16 i + j:  e[16 i + j] = sum_u A[j][u] B[u][i],  A[j][u] = b[256 + j - u],  B[u][i] = Dz[16 i + u],
        // u < 256 (Dz is zero from 256 on).  [16 x 256] x [256 x 32] (16 blocks x {re, im}) = 128 v_mfma_f32_16x16x4_f32, 16 per wave
        // (K split over four wave pairs, one wave of a pair per tile; the partial tiles are summed after the barrier).  The matrix pipe is otherwise idle in this kernel, the
        // B operand is ONE conflict-free ds_read_b32 per MFMA, and all eight waves carry the same load.
        {
            using f32x4 = __attribute__((ext_vector_type(4))) float;
            const int    col = lane & 15, kqm = lane >> 4;
            // !WIN: wave w takes K range u in [32 w, 32 w + 32) of both tiles (8 K-steps x {re, im});
            //  WIN: wave w takes K range [64 (w & 3), +64) of ONE tile (w >> 2: re / im), 16 K-steps -- 16 MFMAs per wave either way
            constexpr int KSW  = 16;
            const int     kw   = wave & 3;
            const float*  pr   = ((wave >> 2) ? Dim : Dre) + 17 * col + kqm + (4 * KSW * 17 / 16) * kw;
            const float*  pa   = hl + 256 + col - kqm - 4 * KSW * kw; // A[j = col][u] = b[256 + j - u], u = 4 KSW kw + 4 i + kqm
            float         av[KSW], br[KSW];
            using e_bf16x8 = __attribute__((ext_vector_type(8))) __bf16;
            e_bf16x8 eb[2][3]; // EBF: B operands: lane (col, kqm) reads Dz[16 col + 64 kw + 32 s + 8 kqm + 0 .. 7] of the re or im planes (tile wave >> 2)
            if constexpr (EBF) {
                const unsigned short* q0 = Dp + (wave >> 2) * 3 * 512 + 16 * col + 64 * kw + 8 * kqm;
#pragma unroll
                for (int s_ = 0; s_ < 2; ++s_)
#pragma unroll
                    for (int p = 0; p < 3; ++p) eb[s_][p] = *reinterpret_cast<const e_bf16x8*>(q0 + p * 512 + 32 * s_);
            } else {
#pragma unroll
                for (int i = 0; i < KSW; ++i) {
                    const int off = 4 * i + (i >> 2); // padded offset of u = 4 KSW kw + 4 i within the window
                    av[i] = FFTONLY ? 0.f : pa[-4 * i];
                    br[i] = FFTONLY ? 0.f : pr[off];
                }
            }
            f32x4 cr = {0.f, 0.f, 0.f, 0.f}, cs = {0.f, 0.f, 0.f, 0.f};
            // One MFMA per ~12-16 butterfly instructions, fenced so that hipcc keeps the order: the wave issues in order, the MFMA
            // occupies the matrix pipe for 32 cycles while the following VALU instructions of the same wave go to the vector pipe.
            // (Left alone hipcc emits the 16 MFMAs back to back in front of the butterflies and the interval grows by their 512 cycles.)
    // EBF: slots 0 .. 11 carry the twelve bf16 MFMAs (two K-steps x {hh, hm, mh -> cr; hl, lh, mm -> cs}), slots 12 .. 15 nothing
#define GR4_MF(i)                                                                           \
    do {                                                                                    \
        if constexpr (EBF) {                                                                \
            if ((i) < 12) { /* (i is a literal or the counter of an unrolled loop: resolved at compile time) */ \
                const int s_ = ((i) / 6) & 1, m_ = (i) % 6;                                 \
                const int pa_ = m_ == 0 || m_ == 1 || m_ == 3 ? 0 : (m_ == 2 || m_ == 5 ? 1 : 2); /* tap plane: hh hm mh hl lh mm */ \
                const int pb_ = m_ == 0 || m_ == 2 || m_ == 4 ? 0 : (m_ == 1 || m_ == 5 ? 1 : 2); /* sample plane */                   \
                __builtin_amdgcn_sched_barrier(0);                                          \
                if (m_ < 3) cr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(e_bf16x8, ea[s_][pa_]), eb[s_][pb_], cr, 0, 0, 0); \
                else cs = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(e_bf16x8, ea[s_][pa_]), eb[s_][pb_], cs, 0, 0, 0);        \
                __builtin_amdgcn_sched_barrier(0);                                          \
            }                                                                               \
        } else if constexpr (!FFTONLY) {                                                    \
            __builtin_amdgcn_sched_barrier(0);                                              \
            cr = __builtin_amdgcn_mfma_f32_16x16x4f32(av[(i)], br[(i)], cr, 0, 0, 0);                 \
            __builtin_amdgcn_sched_barrier(0);                                              \
        }                                                                                   \
    } while (0)
            // ---- X pass B: twiddles W_512^{r k}, 16-point DFT, scatter to S2[c][32 q + k]
#pragma unroll
            for (int g = 0; g < 5; ++g) {
                GR4_MF(g);
#pragma unroll
                for (int r = 3 * g + 1; r < 3 * g + 4; ++r) w[r] = cmul(w[r], WIN ? twBl[r * 32 + kb] : twBr[r]);
            }
#pragma unroll
            for (int n2 = 0; n2 < 4; ++n2) {
                GR4_MF(5 + n2);
                fft16_s1<1>(w, n2);
            }
            GR4_MF(9);
            fft16_twa<1>(w);
            GR4_MF(10);
            fft16_twb<1>(w);
#pragma unroll
            for (int k1 = 0; k1 < 4; ++k1) {
                GR4_MF(11 + k1);
                fft16_s2<1>(w, k1);
            }
            GR4_MF(15);
#pragma unroll
            for (int q = 0; q < 16; ++q) S[cb * kRowB + 32 * q + kb] = w[perm16(q)];
#undef GR4_MF
            // D[row = 4 kqm + r][col] = partial e[16 col + 4 kqm + r]
            // P[K quarter][re, im][256]
            if constexpr (!FFTONLY) *reinterpret_cast<float4*>(P + (wave & 3) * 512 + (wave >> 2) * 256 + 16 * col + 4 * kqm) = make_float4(cr[0] + cs[0], cr[1] + cs[1], cr[2] + cs[2], cr[3] + cs[3]);
        }
        GR4_STAMP(6);
        GR4_LDS_BARRIER(); // #3
        GR4_STAMP(7);
        GR4_DRAIN(5);
        if constexpr (!FFTONLY) { // e = sum of the four partial tiles (fixed order); lane t -> component t >> 8 of e[t & 255]
            const float* pp = P + t;
            reinterpret_cast<float*>(el)[2 * (t & 255) + (t >> 8)] = (pp[0] + pp[512]) + (pp[1024] + pp[1536]);
        }
        // ------------------------------------------------------------------ X pass C (p = 512, radix 16), then H[k] X[k]
        float2 X[16];
        passC(S, X, twCr, t);
        GR4_DRAIN(6);
        [[maybe_unused]] float fr_pk = 0.f; // the largest component of this lane's sixteen bins of X: the frame's peak, for the guard's second condition
        if constexpr (Q4 && !FFTONLY) {
            if (measure) {
#pragma unroll
                for (int q = 0; q < 16; ++q) fr_pk = max3_abs(fr_pk, X[q].x, X[q].y);
            }
        }
#pragma unroll
        for (int q = 0; q < 16; ++q)
            if constexpr (!FFTONLY && !(GR4_FMA_COMBINE && MODE == kModeMag2)) X[perm16(q)] = cmul(Hr[q], X[perm16(q)]); // (headline mode: H enters in the combine, as one fma chain onto E)
        GR4_STAMP(8);
        GR4_LDS_BARRIER(); // #4: every lane has consumed S; e[] is complete
        GR4_STAMP(9);
        GR4_DRAIN(7);

        if constexpr (SPEC) {
#pragma unroll
            for (int q = 0; q < 16; ++q) { pend[q] = X[perm16(q)].x; pendi[q] = X[perm16(q)].y; } // X[t + 512 q], stored during the next frame
        } else if constexpr (FFTONLY) {
#pragma unroll
            for (int q = 0; q < 16; ++q) pend[q] = fmaf(X[perm16(q)].x, X[perm16(q)].x, X[perm16(q)].y * X[perm16(q)].y); // |X[t + 512 q]|^2, stored during the next frame
        } else if constexpr (MODE == kModeMag2) {
            // ------------------------------------------------------------------ E: pass A is a broadcast, then the same passes B and C
#if defined(GR4_T_NOE) // developer timing build (results are wrong): no E transform at all = the bound of 1.0 transforms per frame
#pragma unroll
            for (int q = 0; q < 16; ++q) pend[q] = fmaf(X[perm16(q)].x, X[perm16(q)].x, X[perm16(q)].y * X[perm16(q)].y);
            if (false) {
#elif defined(GR4_T_EINTERP) // developer timing build (results are wrong): E read off a 4 x oversampled 1024-point grid through a 5-tap kernel (the grid's own small transform taken as free,
                             // one barrier standing in for it) = the bound of replacing E's two radix-16 passes by an interpolation
            GR4_LDS_BARRIER();
            {
                float cf[5];
#pragma unroll
                for (int j = 0; j < 5; ++j) cf[j] = twCr[1 + j].x; // (stand-ins for the kernel's five coefficients of this lane's fractional position t & 7)
                const float2* Sg = S + (t >> 3); // (the grid padded by its own first points: no wrap-around arithmetic, every read an immediate offset from one base)
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float2    Eq   = make_float2(0.f, 0.f);
#pragma unroll
                    for (int j = 0; j < 5; ++j) {
                        const float2 g = Sg[64 * q + j];
                        Eq.x = fmaf(cf[j], g.x, Eq.x);
                        Eq.y = fmaf(cf[j], g.y, Eq.y);
                    }
                    const float2 Xq = X[perm16(q)];
                    const float2 Y  = make_float2(fmaf(Hr[q].x, Xq.x, fmaf(-Hr[q].y, Xq.y, Eq.x)), fmaf(Hr[q].x, Xq.y, fmaf(Hr[q].y, Xq.x, Eq.y)));
                    pend[q] = fmaf(Y.x, Y.x, Y.y * Y.y);
                }
            }
            if (false) {
#else
            {
#endif
    #pragma unroll
            for (int r = 0; r < 16; ++r) w[r] = el[cb + 16 * r];
#if defined(GR4_T_NOEB) // developer timing build (results are wrong): E's pass B without its arithmetic = the bound of moving it off the VALU
#pragma unroll
            for (int q = 0; q < 16; ++q) S[cb * kRowB + 32 * q + kb] = w[perm16(q)];
#else
            passB_compute_store(S, w, twBr, cb, kb);
#endif
            GR4_STAMP(10);
#if defined(GR4_T_NOB5) // developer timing build (results are wrong): E's exchange without its barrier = the bound of an E transform with one exchange fewer
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#else
            GR4_LDS_BARRIER(); // #5
#endif
            GR4_STAMP(11);
            passC(S, w, twCr, t);
            GR4_PIN(w);
            GR4_STAMP(12);
            // ------------------------------------------------------------------ FFT(y_f)[k] = H[k] X[k] + E[k];  mag2 = |.|^2  (k = t + 512 q)
    #pragma unroll
            for (int q = 0; q < 16; ++q) {
#if GR4_FMA_COMBINE
                const float2 Xq = X[perm16(q)], Eq = w[perm16(q)]; // H X + E as two fma chains per component (4 instructions instead of 4 + 2)
                const float2 Y  = make_float2(fmaf(Hr[q].x, Xq.x, fmaf(-Hr[q].y, Xq.y, Eq.x)), fmaf(Hr[q].x, Xq.y, fmaf(Hr[q].y, Xq.x, Eq.y)));
#else
                const float2 Y = cadd(X[perm16(q)], w[perm16(q)]);
#endif
                const float m2 = fmaf(Y.x, Y.x, Y.y * Y.y);
                if constexpr (MULTI) { m2sum += m2; m4sum = fmaf(m2, m2, m4sum); } // (the guard's output sums of THIS work item: pend carries the running fold)
                pend[q] = (MULTI && ch > 0) ? pend[q] + m2 : m2; // out[t + 512 q], stored during the next frame (MULTI: math::Add's left fold over the channels)
            }
            }
        } else {
            // ------------------------------------------------------------------ y_f = IFFT(H X) + e through conj(FFT(conj(.))) / N
            // conj(H X) in natural bin order k = t + 512 q IS the pass-A image layout (row k >> 8, column k & 255)
#pragma unroll
            for (int q = 0; q < 16; ++q) S[addrA((t >> 8) + 2 * q, t & 255)] = make_float2(X[perm16(q)].x, -X[perm16(q)].y);
            GR4_LDS_BARRIER();
            GR4_PHASE_FENCE();
            passA_inplace(S, twA, par, n0, sgn);
            GR4_LDS_BARRIER();
            GR4_PHASE_FENCE();
#pragma unroll
            for (int r = 0; r < 16; ++r) w[r] = S[addrA(kb, cb + 16 * r)];
            GR4_LDS_BARRIER();
            GR4_PHASE_FENCE();
            if constexpr (WIN) passB_table_store(S, w, twBl, cb, kb);
            else passB_compute_store(S, w, twBr, cb, kb);
            GR4_LDS_BARRIER();
            GR4_PHASE_FENCE();
            passC(S, w, twCr, t); // w[perm16(q)] = N conj(circular y[n]), n = t + 512 q
            if constexpr (WIN) {
                // ------------------------------------------------------------------ window (w[n] / N) (conj(.) + N e[n]), back to the image layout
                float2 yw[16];
    #pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float2 yv = make_float2(w[perm16(q)].x, -w[perm16(q)].y);
                    if (q == 0 && t < 256) {
                        const float2 ev = el[t];
                        yv.x = fmaf((float)kN, ev.x, yv.x);
                        yv.y = fmaf((float)kN, ev.y, yv.y);
                    }
                    yw[q] = make_float2(yv.x * wr[q], yv.y * wr[q]);
                }
                GR4_LDS_BARRIER(); // every lane has consumed S
                GR4_PHASE_FENCE();
                if constexpr (SMALL) {
                    // ---------------------------------------------------------- 8192 / fftSize frames of fftSize points, each in its padded buffer
                    auto PF = [](int i) { return i + (i >> 5); };
#pragma unroll
                    for (int q = 0; q < 16; ++q) {
                        const int n = t + 512 * q;
                        S[(n >> LOG2NF) * NPF + PF(n & (NF - 1))] = yw[q];
                    }
                    GR4_LDS_BARRIER();
                    const int fl = t / TF, tt = t % TF;
                    float2*   fb = S + fl * NPF;
                    float2    v[16], Xs[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[r] = fb[PF(tt + r * TF)];
                    GR4_LDS_BARRIER(); // everybody holds its first-pass inputs: the buffers may be overwritten
                    float2 b2a = sw2a, b2b = sw2b, b3 = sw3, b3sq = sw3sq; // opaque per iteration (see fft_kernels.hpp: the power chains would be hoisted)
                    asm volatile("" : "+v"(b2a.x), "+v"(b2a.y), "+v"(b2b.x), "+v"(b2b.y), "+v"(b3.x), "+v"(b3.y), "+v"(b3sq.x), "+v"(b3sq.y));
                    fft_small_passes<LOG2NF>(v, fb, tt, b2a, b2b, b3, b3sq, Xs, [] { GR4_LDS_BARRIER(); });
#pragma unroll
                    for (int j = 0; j < 16; ++j) pend[j] = fmaf(Xs[j].x, Xs[j].x, Xs[j].y * Xs[j].y); // frame (8192 / NF) f + fl, bin tt + j TF: stored during the next block
                } else {
    #pragma unroll
                for (int q = 0; q < 16; ++q) S[addrA((t >> 8) + 2 * q, t & 255)] = yw[q];
                GR4_LDS_BARRIER();
                GR4_PHASE_FENCE();
                // ------------------------------------------------------------------ FFT(w y_f), |.|^2
                passA_inplace(S, twA, par, n0, sgn);
                GR4_LDS_BARRIER();
                GR4_PHASE_FENCE();
    #pragma unroll
                for (int r = 0; r < 16; ++r) w[r] = S[addrA(kb, cb + 16 * r)];
                GR4_LDS_BARRIER();
                GR4_PHASE_FENCE();
                passB_table_store(S, w, twBl, cb, kb);
                GR4_LDS_BARRIER();
                GR4_PHASE_FENCE();
                passC(S, w, twCr, t);
    #pragma unroll
                for (int q = 0; q < 16; ++q) pend[q] = fmaf(w[perm16(q)].x, w[perm16(q)].x, w[perm16(q)].y * w[perm16(q)].y); // stored during the next frame
                }
            } else { // kModeFir: y_f[n] = conj(.) / N + e[n], complex, straight to HBM
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    float2 yv = make_float2(w[perm16(q)].x * (1.f / kN), -w[perm16(q)].y * (1.f / kN));
                    if (q == 0 && t < 256) yv = cadd(yv, el[t]);
                    pend[q]  = yv.x; // y_f[t + 512 q], stored during the next frame
                    pendi[q] = yv.y;
                }
            }
        }
        if constexpr (!FFTONLY) {
            if (measure) { // ALL of the lane's outputs: its sixteen bins t + 512 q are a comb over the whole spectrum, any subset of q is not (a low-pass sits in q = 0 and 15)
                float fr_out = MULTI ? m2sum : 0.f;
                [[maybe_unused]] float fr_o4 = MULTI ? m4sum : 0.f; // sum of |Y_k|^4 over this lane's bins
                if constexpr (!MULTI && !Q4) {
#pragma unroll
                    for (int q = 0; q < 16; ++q) fr_out += fmaf(pend[q], pend[q], pendi[q] * pendi[q]);
                }
                if constexpr (!MULTI && Q4) { // two bins per instruction (v_pk_fma_f32 / v_pk_add_f32)
                    using f32x2 = __attribute__((ext_vector_type(2))) float;
                    f32x2 a4 = {0.f, 0.f}, a2 = {0.f, 0.f};
#pragma unroll
                    for (int q = 0; q < 16; q += 2) {
                        const f32x2 pq = {pend[q], pend[q + 1]};
                        a4 = __builtin_elementwise_fma(pq, pq, a4);
                        a2 += pq;
                    }
                    fr_o4  = a4.x + a4.y;
                    fr_out = a2.x + a2.y;
                }
                pw_out += fr_out;
                // the frame's own verdict, per wave (1024 of its points): the wave's sum of out - thr * in on the DPP network (it lands in lane 63; the other lanes keep
                // partial sums nobody reads), its minimum over the frames kept per lane -- no LDS, no barrier, no branch
                if constexpr (Q4) { pw_dprev = fr_o4; pw_iprev = fr_in; pw_kprev = fr_pk; }
                else pw_dprev = fmaf(-a.pw_thr, fr_in, fr_out); // (summed over the wave at the top of the next iteration: the DPP sequence here costs the compiler ten spills)
            }
        }
        fh2 = fh1; fh1 = f; // (the frames of the last two work items: whose verdict sums thread 0 meets one / two iterations later)
        fprev = f;
        if constexpr (MULTI) { // the fold is complete after the last channel; the other work items store nothing
            if (ch != C - 1) fprev = -1;
            if (++ch >= C) { ch = 0; f += fstride; ++fiter; }
        } else {
            f += fstride;
        }
        GR4_STAMP(13);
        GR4_STAMP(14);
    }
    if constexpr (!FFTONLY) {
        if (a.pw != nullptr) { // one pair of atomics per WORKGROUP, spread over 16 slots (4096 atomics on two addresses cost ~15 us at the end of every launch);
                               // the last workgroup to finish folds the slots, hands the totals to the host and re-arms the accumulators
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                pw_in += __shfl_xor(pw_in, off);
                pw_out += __shfl_xor(pw_out, off);
            }
            {   // the last two frames' verdicts
                const float wt = wave_total_lane63(pw_dprev);
                [[maybe_unused]] float wti = 0.f, wtk = 0.f;
                if constexpr (Q4) { wti = wave_total_lane63(pw_iprev); wtk = wave_max_lane63(pw_kprev); }
                if constexpr (!WIN) {
                    float* Gv = reinterpret_cast<float*>(reinterpret_cast<char*>(smem) + kLdsEbfBytes);
                    if ((threadIdx.x & 63) == 63) { Gv[kGvW * (iter & 1) + (threadIdx.x >> 6)] = wt; if constexpr (Q4) { Gv[kGvW * (iter & 1) + 8 + (threadIdx.x >> 6)] = wti; Gv[kGvW * (iter & 1) + 16 + (threadIdx.x >> 6)] = wtk; } }
                } else {
                    if ((threadIdx.x & 63) == 63) { atomicAdd(pw_slot + (iter & 1), wt); if constexpr (Q4) { atomicAdd(pw_slot + 2 + (iter & 1), wti); atomicMax(reinterpret_cast<unsigned*>(pw_slot) + 4 + (iter & 1), __float_as_uint(wtk)); } }
                }
            }
            __syncthreads(); // (P below is not a DMA target; every lane is past its last use of it)
            if ((threadIdx.x & 63) == 0) {
                P[2 * (threadIdx.x >> 6)]     = pw_in;
                P[2 * (threadIdx.x >> 6) + 1] = pw_out;
            }
            __syncthreads();
            if (threadIdx.x == 0) {
                float s0 = 0.f, s1 = 0.f; // the verdict sums by parity of the iteration that filed them: parity iter & 1 = the last frame, the other = the one before it
                if constexpr (!WIN) {
                    const float* Gv = reinterpret_cast<const float*>(reinterpret_cast<const char*>(smem) + kLdsEbfBytes);
#pragma unroll
                    for (int w = 0; w < 8; ++w) { s0 += Gv[w]; s1 += Gv[kGvW + w]; }
                    if constexpr (Q4) {
                        float i0 = 0.f, i1 = 0.f, k0 = 0.f, k1 = 0.f;
#pragma unroll
                        for (int w = 0; w < 8; ++w) { i0 += Gv[8 + w]; i1 += Gv[kGvW + 8 + w]; k0 = __uint_as_float(max(__float_as_uint(k0), __float_as_uint(Gv[16 + w]))); k1 = __uint_as_float(max(__float_as_uint(k1), __float_as_uint(Gv[kGvW + 16 + w]))); }
                        s0 = verdict(s0, i0, k0); // (a slot nothing was filed in holds zeros: verdict 0, not marked)
                        s1 = verdict(s1, i1, k1);
                    }
                    pw_dmin = fminf(pw_dmin, fminf(s0, s1));
                    pw_nmark += (unsigned)(s0 < 0.f) + (unsigned)(s1 < 0.f);
                } else {
                    s0 = atomicExch(pw_slot, 0.f); s1 = atomicExch(pw_slot + 1, 0.f); // (the slots are zero again for the next launch)
                    if constexpr (Q4) {
                        const float i0 = atomicExch(pw_slot + 2, 0.f), i1 = atomicExch(pw_slot + 3, 0.f);
                        s0 = verdict(s0, i0, __uint_as_float(atomicExch(reinterpret_cast<unsigned*>(pw_slot) + 4, 0u)));
                        s1 = verdict(s1, i1, __uint_as_float(atomicExch(reinterpret_cast<unsigned*>(pw_slot) + 5, 0u)));
                    }
                    pw_dmin = fminf(fminf(pw_dmin, pw_pend), fminf(s0, s1));
                    pw_nmark += (unsigned)(pw_pend < 0.f) + (unsigned)(s0 < 0.f) + (unsigned)(s1 < 0.f);
                }
                if (a.fflags != nullptr) { // the last two work items: fh1 the last, fh2 the one before it
                    const float last = (iter & 1) ? s1 : s0, before = (iter & 1) ? s0 : s1;
                    if constexpr (MULTI) {
                        if (fh1 >= 0 && last < 0.f) a.fflags[fh1] = 1;
                        if (fh2 >= 0 && before < 0.f) a.fflags[fh2] = 1;
                    } else {
                        if (fh1 >= 0) a.fflags[fh1] = last < 0.f ? 1 : 0;
                        if (fh2 >= 0) a.fflags[fh2] = before < 0.f ? 1 : 0;
                    }
                }
                if (pw_dmin < 0.f) atomicAdd(reinterpret_cast<unsigned*>(a.pw + 33), pw_nmark ? pw_nmark : 1u); // (word 33: how many frames of this launch were marked)
                float si = 0.f, so = 0.f;
#pragma unroll
                for (int w = 0; w < kT / 64; ++w) { si += P[2 * w]; so += P[2 * w + 1]; }
                float* slot = a.pw + 2 * (gslot & 15);
                atomicAdd(slot, si);
                atomicAdd(slot + 1, so);
                __threadfence();
                unsigned* done = reinterpret_cast<unsigned*>(a.pw + 32);
                if (atomicAdd(done, 1u) == gcount - 1) {
                    __threadfence();
                    float tin = 0.f, tout = 0.f;
                    for (int k = 0; k < 16; ++k) { tin += atomicExch(a.pw + 2 * k, 0.f); tout += atomicExch(a.pw + 2 * k + 1, 0.f); }
                    atomicExch(done, 0u);
                    const unsigned nm = atomicExch(reinterpret_cast<unsigned*>(a.pw + 33), 0u);
                    reinterpret_cast<volatile unsigned*>(a.pw_host)[3] = nm; // how many frames of this launch were marked
                    if (a.no_tier && nm) atomicAdd_system(reinterpret_cast<unsigned*>(a.pw_host) + 4, nm); // no 22-bit tier behind this launch: every one of them is a float64 frame (the running total chain_td16_kernel adds to otherwise)
                    // ONE 8-byte store: the pair arrives whole; then, behind a system-scope fence, the launch's sequence number -- what a waiting host spins on
                    // (a few microseconds after the last workgroup instead of a stream synchronisation's wake-up)
                    *reinterpret_cast<volatile unsigned long long*>(a.pw_host) = (unsigned long long)__float_as_uint(tin) | ((unsigned long long)__float_as_uint(tout) << 32);
                    __threadfence_system();
                    reinterpret_cast<volatile unsigned*>(a.pw_host)[2] = a.pw_seq;
                }
            }
        }
    }
    if (DEFER && fprev >= 0) {
        const rsrc_t rq = make_rsrc(a.out + fprev * kN * (FIR ? 2 : 1), kN * sizeof(float) * (FIR ? 2 : 1));
#pragma unroll
        for (int q = 0; q < 16; ++q) {
            if constexpr (FIR) buf_store_f2(rq, make_float2(pend[q], pendi[q]), t0 * 8, q * 4096);
            else if constexpr (SMALL) buf_store_f(rq, pend[q], ((t0 / TF) * NF + t0 % TF) * 4, q * TF * 4);
            else buf_store_f(rq, pend[q], t0 * 4, q * 2048);
        }
    }
#undef GR4_DRAIN
}
template <int MODE, int LOG2NF = 13>
__global__ __launch_bounds__(kT, 2) void chain_fd_kernel(ChainFdArgs a) { chain_fd_body<MODE, LOG2NF, false>(a, nullptr); }

// ---- the guard's second evaluation, on the device: the frames chain_fd_kernel marked in a.fflags (output power below the threshold x input power: the fast
// convolution's error, relative to the INPUT, shows against such an output) again with the filter in the TIME domain -- y_f = sum_k b[k] x[n - k] on the FP64 matrix
// pipe (the band form of fir_exact.hip: float32 x float32 products exact, float64 sums, one rounding), x window, the frame transform of this file from LDS, |.|^2 over
// the fused kernel's output.  Launched behind every guarded launch; a workgroup whose frames are unmarked reads their flag bytes and leaves (an ordinary stream costs
// one near-empty launch).  No host in the loop: gr4hip_chain_process returns when both launches are enqueued (until round 5 it waited for the verdict and redid the
// whole span on the time-domain kernel pair).  A frame whose staged window holds a non-finite sample keeps the fused kernel's result.
// LDS: the frame image S (kSLen float2) + the staged window as two padded float planes (8192 + 256 samples each, 4 floats of padding per 16) + the tap row.
constexpr int kRdHb = 256, kRdKw = 272, kRdL = kN + kRdHb, kRdLp = kRdL + 4 * (kRdL / 16) + 4;
constexpr size_t kRdLdsBytes = (size_t)kSLen * sizeof(float2) + (size_t)(2 * kRdLp + kRdKw + 16) * sizeof(float);
static_assert(kRdLdsBytes <= 160 * 1024, "LDS budget of one CU");
// FOLD: several channels of a launch that kept only sum_c |FFT(fir(x_c))|^2 (gr4hip_chain_process_multi with shared taps): a marked frame is evaluated again for EVERY channel
// (mc->xs / mc->hists), the fold kept in registers as the fused kernel keeps it (math::Add's left fold over the inputs); a non-finite sample in any channel's window leaves the
// frame as it is.
template <int MODE, int LOG2NF, bool FOLD>
__device__ __forceinline__ void chain_redo_body(ChainFdArgs& a, int ntaps, const ChainFdMulti* mc) {
    constexpr bool WIN = MODE != kModeMag2, SMALL = MODE == kModeWinSmall;
    static_assert(!FOLD || MODE == kModeMag2, "the fold: 8192-point rectangular-window chains only");
    using f64x4_r = __attribute__((ext_vector_type(4))) double;
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2* S  = smem;
    float*  sg = reinterpret_cast<float*>(smem + kSLen); // [2][kRdLp]
    float*  tz = sg + 2 * kRdLp;                         // tz[15 + k] = b[k], zeros either side
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, col = lane & 15, kq = lane >> 4;
    auto ph = [](int i) { return i + 4 * (i >> 4); };
    const int  fpb    = a.redo_flags_per_block > 0 ? a.redo_flags_per_block : 1, hlen = a.redo_hist_len > 0 ? a.redo_hist_len : 256;
    const long n_span = a.redo_n > 0 ? a.redo_n : a.n_frames * kN, n_flags = (n_span + kN / fpb - 1) / (kN / fpb);
    const int  min_flag = a.redo_min_flag > 0 ? a.redo_min_flag : 1;
    const auto marked = [&](long f) { // any flag byte of block f (at or above the level this launch answers to)
        int m = 0;
        for (long b = f * fpb; b < (f + 1) * fpb && b < n_flags; ++b) m |= (int)(a.fflags[b] >= min_flag);
        return m;
    };
    {   // an ordinary stream marks nothing: every lane looks at its share of this workgroup's flag bytes at once, and the workgroup leaves (measured: a lane that walks its
        // 512 bytes one dependent load after the other costs the headline launch 0.35 ms)
        int any = 0;
        for (long f = (long)blockIdx.x + (long)t * gridDim.x; f < a.n_frames; f += (long)kT * gridDim.x) any |= marked(f);
        if (!__syncthreads_or(any)) return;
    }
    for (int i = t; i < kRdKw + 15; i += kT) { const int k = i - 15; tz[i] = (k >= 0 && k < ntaps) ? a.taps[k] : 0.f; }
    const int   n0 = t >> 1, par = t & 1;
    const float sgn = par ? -1.f : 1.f;
    const int   cb = lane & 15;
    const int   kb = 2 * wave + (kq >> 1) + 16 * (kq & 1);
    float2 twBr[16], twCr[16], twA[16];
#pragma unroll
    for (int r = 1; r < 16; ++r) { twBr[r] = a.twB[r * 32 + kb]; twCr[r] = a.twC[r * 512 + t]; }
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) twA[k1] = (t & 1) ? w32(k1) : make_float2(1.f, 0.f);
    constexpr int NF = 1 << (SMALL ? LOG2NF : 8), TF = NF / 16, NPF = NF + NF / 32;
    float2        sw2a = make_float2(1.f, 0.f), sw2b = sw2a, sw3 = sw2a, sw3sq = sw2a;
    if constexpr (SMALL) {
        const int tt = t % TF;
        sw2a  = a.twS[(tt & 15) * (NF / 256)];
        sw2b  = a.twS[2 * (tt & 15) * (NF / 256)];
        sw3   = a.twS[tt & 255];
        sw3sq = a.twS[(2 * (tt & 255)) & (NF - 1)];
    }
    const int comp = wave & 1;
    const int C = FOLD ? mc->n_ch : 1;
    for (long f = blockIdx.x; f < a.n_frames; f += gridDim.x) {
        if (marked(f) == 0) continue; // (uniform)
        [[maybe_unused]] float accm[16]; // FOLD: the running sum over the channels, bins t + 512 q
        [[maybe_unused]] bool  spoilt = false;
      for (int ch = 0; ch < C; ++ch) {
        const float2* xc = a.x;
        const float2* hc = a.hist;
        if constexpr (FOLD) { xc = mc->xs[ch]; hc = mc->hists[ch]; }
        __syncthreads();              // the previous frame's readers are done with S and the planes
        int nf = 0;
        for (int i = t; i < kRdL; i += kT) {
            const long   sidx = f * kN - kRdHb + i;
            const float2 v = sidx >= 0 ? (sidx < n_span ? xc[sidx] : make_float2(0.f, 0.f)) : (sidx >= -(long)hlen ? hc[hlen + sidx] : make_float2(0.f, 0.f));
            nf |= (int)!(__builtin_fabsf(v.x) <= 3.4028234663852886e38f) | (int)!(__builtin_fabsf(v.y) <= 3.4028234663852886e38f);
            sg[ph(i)]         = v.x;
            sg[kRdLp + ph(i)] = v.y;
        }
        if (__syncthreads_or(nf)) { spoilt = true; break; } // (uniform; single channel: the fused kernel's result stands)
        // ---- y_f on the FP64 matrix pipe: wave = component x (tile rows (wave >> 1) + 4 i); a tile row = 256 consecutive outputs, D[row = kq + 4 r][col] = output 256 tr + 16 col + kq + 4 r
        for (int i = 0; i < 8; ++i) {
            const int    tr = (wave >> 1) + 4 * i;
            const float* sb = sg + comp * kRdLp + (256 * tr + 16 * col) + 4 * (16 * tr + col) + kq;
            const float* ta = tz + 15 + kRdHb + col - kq;
            f64x4_r      acc = {0., 0., 0., 0.}, acc1 = {0., 0., 0., 0.}; // (two accumulators over alternating K-steps: one dependent chain per wave leaves the pipe idle for its latency)
            for (int k0 = 0, pad = 0; k0 < kRdKw / 4; k0 += 4, pad += 4) { // a pad of 4 floats every 16 samples = every 4 K-steps of 4
                float av[4], bv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { av[q] = ta[-4 * (k0 + q)]; bv[q] = sb[4 * (k0 + q) + pad]; }
                acc  = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[0], (double)bv[0], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[1], (double)bv[1], acc1, 0, 0, 0);
                acc  = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[2], (double)bv[2], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[3], (double)bv[3], acc1, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[r] += acc1[r];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int o = 256 * tr + 16 * col + kq + 4 * r;
                float     v = (float)acc[r];
                if constexpr (WIN) v *= a.win[o] * (float)kN; // (the table holds window / N for the fused kernel's unnormalised inverse transform)
                if constexpr (SMALL) reinterpret_cast<float*>(S)[2 * ((o >> LOG2NF) * NPF + (o & (NF - 1)) + ((o & (NF - 1)) >> 5)) + comp] = v;
                else reinterpret_cast<float*>(S)[2 * addrA(o >> 8, o & 255) + comp] = v;
            }
        }
        __syncthreads();
        float* out = a.out + f * kN;
        if constexpr (SMALL) {
            auto PF = [](int i) { return i + (i >> 5); };
            const int fl = t / TF, tt = t % TF;
            float2*   fb = S + fl * NPF;
            float2    v[16], Xs[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fb[PF(tt + r * TF)];
            __syncthreads();
            float2 b2a = sw2a, b2b = sw2b, b3 = sw3, b3sq = sw3sq;
            asm volatile("" : "+v"(b2a.x), "+v"(b2a.y), "+v"(b2b.x), "+v"(b2b.y), "+v"(b3.x), "+v"(b3.y), "+v"(b3sq.x), "+v"(b3sq.y));
            fft_small_passes<LOG2NF>(v, fb, tt, b2a, b2b, b3, b3sq, Xs, [] { __syncthreads(); });
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int off = (t / TF) * NF + t % TF + j * TF;
                if (f * kN + off < n_span) out[off] = fmaf(Xs[j].x, Xs[j].x, Xs[j].y * Xs[j].y);
            }
        } else {
            passA_inplace(S, twA, par, n0, sgn);
            __syncthreads();
            float2 w[16], X[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) w[r] = S[addrA(kb, cb + 16 * r)];
            __syncthreads();
            passB_compute_store(S, w, twBr, cb, kb);
            __syncthreads();
            passC(S, X, twCr, t);
            if constexpr (FOLD) {
#pragma unroll
                for (int q = 0; q < 16; ++q) {
                    const float m2 = fmaf(X[perm16(q)].x, X[perm16(q)].x, X[perm16(q)].y * X[perm16(q)].y);
                    accm[q] = ch == 0 ? m2 : accm[q] + m2;
                }
            } else {
#pragma unroll
                for (int q = 0; q < 16; ++q) out[t + 512 * q] = fmaf(X[perm16(q)].x, X[perm16(q)].x, X[perm16(q)].y * X[perm16(q)].y);
            }
        }
      } // channels
        if constexpr (FOLD) {
            if (!spoilt) {
                float* out = a.out + f * kN;
#pragma unroll
                for (int q = 0; q < 16; ++q) out[t + 512 * q] = accm[q];
            }
        }
    }
}
// ---------------------------------------------------------------------------------------------------------------------------------------
// chain_td16_kernel (round 6): the chain in the TIME domain at the headline shape -- <= 256 taps, 8192-sample blocks, any window, fftSize 256 .. 8192 -- for the frames the
// fused fast convolution has marked (flag byte 1), or for every frame of a stream the guard has moved (the host sets every byte to 1):
//     y = sum_k b[k] x[n - k] on the f16 matrix pipe (two-term splits under ONE block exponent per frame, three products per tap: fir_f16.hip's arithmetic and tile map,
//     four 2048-output segments per frame, two per half of the workgroup)  ->  x window  ->  the frame transform of this file from LDS  ->  |.|^2.
// Its products carry 22 bits: the error is ~1.3e-7 of the PRODUCTS' level sqrt(sum b^2) rms(x) -- relative to the filtered output while the filter passes what white noise
// would pass, which is exactly the stream the fast convolution (error relative to the INPUT) cannot serve: a narrow channel filter over wide-band noise.  A frame whose
// filter output lies more than 15 dB below that level (a strong rejected interferer; chain.hip kChainPairGuardRatio) gets flag 2 and is left to chain_redo_kernel's float64
// products behind this launch, like a frame with a non-finite or absurdly ranged sample.  Until round 6 EVERY marked frame took the float64 evaluation: 31 Gsamples/s on a
// stream that marks all of them (34 = 45 % of the FP64 matrix pipe's peak).
constexpr int kT16KS = 9, kT16Kw = 32 * kT16KS, kT16Hb = kT16Kw - 16;                     // window of 288 samples: Hb = 272 in front of a 16-output tile
constexpr int kT16NS = kN + kT16Hb;                                                         // staged complex samples per frame
constexpr int kT16PL = kT16NS + 8 * (kT16NS / 128 + 1) + 16;                                // f16 elements per plane (16 bytes of padding per 128 samples)
constexpr int kT16NL4 = (kT16NS / 2 + kT - 1) / kT;                                         // float4 (two complex samples) a lane stages
constexpr size_t kT16LdsBytes = (size_t)kSLen * sizeof(float2) + (size_t)4 * kT16PL * sizeof(unsigned short) + 64 * sizeof(float);
static_assert(kT16LdsBytes <= 160 * 1024, "LDS budget of one CU");
static_assert(kT16NS % 16 == 0, "whole 16-sample groups");

#ifdef GR4_FD_TIMING
#define GR4_T16_STAMP(i, fr)                                                                          \
    do {                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                            \
        unsigned long long t_;                                                                        \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory");                   \
        if ((threadIdx.x & 63) == 0 && a.dbg) a.dbg[((fr) * 8 + (threadIdx.x >> 6)) * 16 + (i)] = t_; \
        __builtin_amdgcn_sched_barrier(0);                                                            \
    } while (0)
#else
#define GR4_T16_STAMP(i, fr) do { } while (0)
#endif
template <int MODE, int LOG2NF>
__device__ __forceinline__ void chain_td16_body(ChainFdArgs& a) {
    constexpr bool WIN = MODE != kModeMag2, SMALL = MODE == kModeWinSmall;
    constexpr int  KS = kT16KS, Hb = kT16Hb, NS = kT16NS, PL = kT16PL, NL4 = kT16NL4;
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2*         S   = smem;
    unsigned short* pls = reinterpret_cast<unsigned short*>(smem + kSLen); // planes re1, re2, im1, im2
    unsigned*       st  = reinterpret_cast<unsigned*>(pls + 4 * PL);       // [0..7] largest magnitude per wave, [8..15] quietest group, [16..23] input power, [24..31] output power
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, col = lane & 15, kq = lane >> 4;
    auto P = [](int s_) { return s_ + 8 * (s_ >> 7); };
    {   // nothing marked (the ordinary stream): leave at once -- every lane looks at its share of the flag bytes
        int any = 0;
        for (long f = (long)blockIdx.x + (long)t * gridDim.x; f < a.n_frames; f += (long)kT * gridDim.x) any |= (int)(a.fflags[f] == 1);
        if (!__syncthreads_or(any)) return;
    }
    const unsigned short* blk   = a.hfrag;
    const u32x4_h*        afrag = reinterpret_cast<const u32x4_h*>(blk);
    const float           inv_t = *reinterpret_cast<const float*>(blk + KS * 1536);
    const float           h2_128 = *reinterpret_cast<const float*>(blk + KS * 1536 + 4); // sum b^2 / 128
    const float           gthr  = h2_128 * 128.f * a.td16_thr;
    // transform tables (as chain_redo_kernel)
    const int   n0 = t >> 1, par = t & 1;
    const float sgn = par ? -1.f : 1.f;
    const int   cb = lane & 15;
    const int   kb = 2 * wave + (kq >> 1) + 16 * (kq & 1);
    constexpr int NF = 1 << (SMALL ? LOG2NF : 8), TF = NF / 16, NPF = NF + NF / 32;
    const long  n_span = a.n_frames * kN;
    const int   hlen = 256;
    // this workgroup's next frame marked 1 at or behind f (uniform)
    const auto next_marked = [&](long f) {
        while (f < a.n_frames && a.fflags[f] != 1) f += gridDim.x;
        return f;
    };
    // NS samples from x[f N - Hb ...] into registers (the 256 samples of carried history in front of the span, zeros further back: they meet zero taps)
    float4 v4[NL4];
    const auto load_frame = [&](long f) {
        if (f > 0) { // (uniform) the staged range starts inside the span: one buffer descriptor, no per-lane address or bound (a read behind the span's end returns zeros)
            const long   i0   = f * kN - Hb;
            const long   nrec = n_span - i0 < (long)NS ? n_span - i0 : (long)NS;
            const rsrc_t r    = make_rsrc(a.x + i0, (unsigned)(nrec * 8));
#pragma unroll
            for (int u = 0; u < NL4; ++u) {
                const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, t * 16, kT * u * 16, 0);
                v4[u]        = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
            }
            return;
        }
        // the span's first frame: the 256 samples of carried history in front of it, zeros further back.  Two descriptors, no branch: a lane's pair lies in the span, in the
        // history or in front of both, and a buffer read outside its descriptor (a negative offset is a huge unsigned one) returns zeros -- so the two reads are added
        const long   nrec = n_span < (long)kN ? n_span : (long)kN;
        const rsrc_t rx = make_rsrc(a.x, (unsigned)(nrec * 8)), rh = make_rsrc(a.hist, (unsigned)(hlen * 8));
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const int  i  = 2 * (t + kT * u) - Hb; // (even, like Hb and the history's length: a pair never straddles a boundary)
            const auto vx = __builtin_amdgcn_raw_buffer_load_b128(rx, i * 8, 0, 0), vh = __builtin_amdgcn_raw_buffer_load_b128(rh, (hlen + i) * 8, 0, 0);
            v4[u] = make_float4(__uint_as_float(vx[0] | vh[0]), __uint_as_float(vx[1] | vh[1]), __uint_as_float(vx[2] | vh[2]), __uint_as_float(vx[3] | vh[3]));
        }
    };
    unsigned n_esc = 0; // frames this workgroup left to the float64 evaluation
    long f = next_marked(blockIdx.x);
    if (f < a.n_frames) load_frame(f);
    for (; f < a.n_frames;) {
        const long fnext = next_marked(f + gridDim.x);
        __syncthreads();                // the previous frame's readers are done with S, the planes and the statistics
        GR4_T16_STAMP(0, f);
        // ---- statistics of the staged samples (one block exponent per frame)
        float  mf = 0.f, px = 0.f;
        unsigned mn = 0xffffffffu;
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const float4 s = v4[u];
            const float m4 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(s.x), __builtin_fabsf(s.y)), __builtin_fmaxf(__builtin_fabsf(s.z), __builtin_fabsf(s.w)));
            mf = __builtin_fmaxf(mf, m4);
            mn = min(mn, __float_as_uint(m4) - 1u);
            px = fmaf(s.x, s.x, fmaf(s.y, s.y, fmaf(s.z, s.z, fmaf(s.w, s.w, px))));
        }
        {
            unsigned mx = __float_as_uint(mf);
            mx = hf_wave_reduce_u32(mx, [](unsigned a_, unsigned b_) { return a_ > b_ ? a_ : b_; });
            mn = hf_wave_reduce_u32(mn, [](unsigned a_, unsigned b_) { return a_ < b_ ? a_ : b_; });
            px = hf_wave_sum(px);
            if (lane == 0) { st[wave] = mx; st[8 + wave] = mn; st[16 + wave] = __float_as_uint(px); }
        }
        __syncthreads();
        GR4_T16_STAMP(1, f);
        unsigned mx = 0, mq = 0xffffffffu;
        px = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) { mx = max(mx, st[w]); mq = min(mq, st[8 + w]); px += __uint_as_float(st[16 + w]); }
        const int e = (int)(mx >> 23), el = (int)(mq >> 23);
        // a non-finite sample, a spread beyond what one block exponent carries, powers outside float32's range: the float64 evaluation's frame (it leaves a non-finite one as the fused kernel wrote it)
        const bool hard = e == 255 || px != px || (mq != 0xffffffffu && e - el > kHfMaxRange) || (mx != 0u && (e < 127 - 60 || e > 127 + 60));
        if (hard) { if (t == 0) a.fflags[f] = 2; ++n_esc; f = fnext; if (f < a.n_frames) load_frame(f); continue; } // (uniform)
        const int   ec = e < 15 ? 15 : (e > 254 ? 254 : e);
        const float s_cur = __uint_as_float((unsigned)(268 - ec) << 23), inv_cur = __uint_as_float((unsigned)(ec - 14) << 23);
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const int q = t + kT * u;
            unsigned rh, rl, ih, il;
            hf_split2(v4[u].x, v4[u].z, s_cur, rh, rl);
            hf_split2(v4[u].y, v4[u].w, s_cur, ih, il);
            const int el_ = q < NS / 2 ? P(2 * q) : PL - 16 + 2 * (lane & 7); // (lanes past the staged range: the spare elements behind each plane)
            *reinterpret_cast<unsigned*>(pls + el_)          = rh;
            *reinterpret_cast<unsigned*>(pls + PL + el_)     = rl;
            *reinterpret_cast<unsigned*>(pls + 2 * PL + el_) = ih;
            *reinterpret_cast<unsigned*>(pls + 3 * PL + el_) = il;
        }
        __syncthreads();
        const long fcur = f;
        GR4_T16_STAMP(2, fcur);
        // ---- y on the f16 matrix pipe, x window, into the transform's image
        u32x4_h        af[2][KS];
        const u32x4_h* afp = afrag;
        asm volatile("" : "+s"(afp)); // (the fragments are the same for every frame: loaded here, per frame, so that their 72 registers are free during the transform -- hoisted out of the
                                      // frame loop they pushed the kernel into scratch)
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) af[p][ks] = afp[(p * KS + ks) * 64 + lane];
        const int   ek  = (int)((__float_as_uint(inv_t) >> 23) & 255) + (int)((__float_as_uint(inv_cur) >> 23) & 255) - 254;
        const bool  one = ek > -120 && ek < 120;
        const float k1 = one ? inv_t * inv_cur : inv_t, k2 = one ? 1.f : inv_cur, kd = k1 * (1.f / 2048.f);
        float py = 0.f;
#ifndef GR4_T16_NT
#define GR4_T16_NT 2   // tiles of a 128-output column per wave: 2 (fir_f16.hip's map: four waves per 2048-output segment, two passes over the frame) or 4 (two waves per segment, one pass)
#endif
#ifndef GR4_T16_PIPE
#define GR4_T16_PIPE 0 // 1: the sample fragments of step m + 1 are read before the products of step m are issued
#endif
        // Measured (profiles/r06_chain_td16.txt; settled stream, Gsamples/s rectangular / Hann): NT 2 144 / 131; NT 2 + PIPE 149 / 120; NT 4 + PIPE 126 / 94.  One stream of sample
        // fragments serves a wave's NT tiles (tile jj meets fragment m at K-step m - jj), so NT 4 reads 40 % less LDS per product and PIPE hides the reads' queueing behind the
        // wave's own products -- the products' phase does shrink (16.8k -> 14.5k -> 13.4k cycles per frame) -- but 64 accumulator + 72 tap-fragment + 32 sample-fragment registers
        // at 256 per lane put the staging and the transform into scratch, which costs more than the products gain.
        constexpr int NT = GR4_T16_NT, NMS = KS + NT - 1;
#pragma unroll 1
        for (int pass = 0; pass < 4 / NT; ++pass) {
            const int sg  = NT == 4 ? wave >> 1 : (wave >> 2) + 2 * pass;
            const int tb2 = NT == 4 ? wave & 1 : ((wave & 3) >> 1) + 4 * (wave & 1); // NT 2: waves 0, 1 of a group: tiles {0, 2} / {4, 6}; waves 2, 3: tiles {1, 3} / {5, 7}
            const int sb  = 2048 * sg + 128 * col + 16 * tb2 + 8 * kq;
            f32x4_h c[2 * NT], d[2 * NT]; // index 2 tile + component
#pragma unroll
            for (int j = 0; j < 2 * NT; ++j) c[j] = d[j] = f32x4_h{0.f, 0.f, 0.f, 0.f};
            f16x8_h bq[2][4];
            const auto frag = [&](int m, f16x8_h (&b)[4]) {
                const unsigned short* q = pls + P(sb + 32 * m);
#pragma unroll
                for (int k = 0; k < 4; ++k) b[k] = *reinterpret_cast<const f16x8_h*>(q + k * PL); // re1, re2, im1, im2
            };
#ifndef GR4_T16_NOFIR
            if (GR4_T16_PIPE) frag(0, bq[0]);
#pragma unroll
            for (int m = 0; m < NMS; ++m) {
                if (GR4_T16_PIPE) {
                    if (m + 1 < NMS) frag(m + 1, bq[(m + 1) & 1]);
                    __builtin_amdgcn_sched_barrier(0);
                } else frag(m, bq[m & 1]);
                const f16x8_h b1[2] = {bq[m & 1][0], bq[m & 1][2]}, b2[2] = {bq[m & 1][1], bq[m & 1][3]};
#pragma unroll
                for (int jj = 0; jj < NT; ++jj) {
                    const int ks = m - jj;
                    if (ks < 0 || ks >= KS) continue;
                    const f16x8_h a1 = __builtin_bit_cast(f16x8_h, af[0][ks]), a2 = __builtin_bit_cast(f16x8_h, af[1][ks]);
#pragma unroll
                    for (int cp = 0; cp < 2; ++cp) {
                        c[2 * jj + cp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b1[cp], c[2 * jj + cp], 0, 0, 0);
                        d[2 * jj + cp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1, b2[cp], d[2 * jj + cp], 0, 0, 0);
                        d[2 * jj + cp] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b1[cp], d[2 * jj + cp], 0, 0, 0);
                    }
                }
            }
#endif
#pragma unroll
            for (int jj = 0; jj < NT; ++jj) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    float vr = fmaf(d[2 * jj][r], kd, c[2 * jj][r] * k1), vi = fmaf(d[2 * jj + 1][r], kd, c[2 * jj + 1][r] * k1);
                    if (!one) { vr *= k2; vi *= k2; }
                    py = fmaf(vr, vr, fmaf(vi, vi, py));
                    const int o = 2048 * sg + 128 * col + 16 * (tb2 + 2 * jj) + 4 * kq + r;
                    if constexpr (WIN) { const float w = a.win[o] * (float)kN; vr *= w; vi *= w; } // (the table holds window / N for the fused kernel's unnormalised inverse transform)
                    if constexpr (SMALL) S[(o >> LOG2NF) * NPF + (o & (NF - 1)) + ((o & (NF - 1)) >> 5)] = make_float2(vr, vi);
                    else S[addrA(o >> 8, o & 255)] = make_float2(vr, vi);
                }
            }
        }
        GR4_T16_STAMP(3, fcur);
        py = hf_wave_sum(py);
        if (lane == 0) st[24 + wave] = __float_as_uint(py);
        f = fnext;
        __syncthreads();
        GR4_T16_STAMP(4, fcur);
        py = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) py += __uint_as_float(st[24 + w]);
#ifdef GR4_T16_NOFFT // timing-only build: no transform, no result
        if (py != 12345.f) { if (f < a.n_frames) load_frame(f); continue; }
#endif
        if (!(py >= gthr * px)) { if (t == 0) a.fflags[fcur] = 2; ++n_esc; if (f < a.n_frames) load_frame(f); continue; } // (uniform) the filter removes too much of what it is given for 22-bit products: float64 behind this launch
        // ---- the frame transform and |.|^2 (chain_redo_kernel's)
        float* out = a.out + fcur * kN;
        if constexpr (SMALL) {
            float2 sw2a, sw2b, sw3, sw3sq;
            {
                const int     tt  = t % TF;
                const float2* tws = a.twS;
                asm volatile("" : "+s"(tws)); // (per frame, like the fragments)
                sw2a  = tws[(tt & 15) * (NF / 256)];
                sw2b  = tws[2 * (tt & 15) * (NF / 256)];
                sw3   = tws[tt & 255];
                sw3sq = tws[(2 * (tt & 255)) & (NF - 1)];
            }
            auto PF = [](int i) { return i + (i >> 5); };
            const int fl = t / TF, tt = t % TF;
            float2*   fb = S + fl * NPF;
            float2    v[16], Xs[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = fb[PF(tt + r * TF)];
            __syncthreads();
            fft_small_passes<LOG2NF>(v, fb, tt, sw2a, sw2b, sw3, sw3sq, Xs, [] { __syncthreads(); });
            if (f < a.n_frames) load_frame(f); // the next frame's samples are on their way while this one's spectra leave
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int off = (t / TF) * NF + t % TF + j * TF;
                out[off] = fmaf(Xs[j].x, Xs[j].x, Xs[j].y * Xs[j].y);
            }
        } else {
            float2 twA[16];
            int    odd = t & 1;
            asm volatile("" : "+v"(odd)); // (per frame, like the fragments: 32 registers)
#pragma unroll
            for (int k1_ = 0; k1_ < 16; ++k1_) twA[k1_] = odd ? w32(k1_) : make_float2(1.f, 0.f);
            passA_inplace(S, twA, par, n0, sgn);
            __syncthreads();
            GR4_T16_STAMP(5, fcur);
            if (f < a.n_frames) load_frame(f); // the next frame's samples are on their way during passes B and C (~3 us); behind pass A, whose 32 twiddle registers are free again
            float2 w[16], X[16], twr[16];
            const float2 *twb = a.twB, *twc = a.twC;
            asm volatile("" : "+s"(twb), "+s"(twc)); // (per frame, like the fragments)
#pragma unroll
            for (int r = 0; r < 16; ++r) w[r] = S[addrA(kb, cb + 16 * r)];
#pragma unroll
            for (int r = 1; r < 16; ++r) twr[r] = twb[r * 32 + kb];
            __syncthreads();
            passB_compute_store(S, w, twr, cb, kb);
#pragma unroll
            for (int r = 1; r < 16; ++r) twr[r] = twc[r * 512 + t];
            __syncthreads();
            GR4_T16_STAMP(6, fcur);
            passC(S, X, twr, t);
            // ---- the second opinion: this frame's spectrum from the fused launch is still in `out`.  The two evaluations err in different ways -- the fast convolution by K sqrt(R4)
            // spread over the bins, the 22-bit products COHERENTLY where a tone is rejected (its residue is off by 2^-22 sum|b| / |H(f)| of itself: 4.7e-5 measured on a 17-tap filter
            // 51 dB down, tools/fuzz_chain.py wide) -- so where they agree within kTd16Agree of max(value, rms of the frame's |Y|^2) in EVERY bin both are right and this one is stored;
            // where they do not, nobody knows which, and the frame goes to the float64 evaluation
            float m2[16], s4 = 0.f;
#pragma unroll
            for (int q = 0; q < 16; ++q) { m2[q] = fmaf(X[perm16(q)].x, X[perm16(q)].x, X[perm16(q)].y * X[perm16(q)].y); s4 = fmaf(m2[q], m2[q], s4); }
            s4 = hf_wave_sum(s4);
            if (lane == 0) st[32 + wave] = __float_as_uint(s4);
            __syncthreads();
            s4 = 0.f;
#pragma unroll
            for (int w_ = 0; w_ < 8; ++w_) s4 += __uint_as_float(st[32 + w_]);
            const float l2 = __builtin_sqrtf(s4 * (1.f / (float)kN)); // rms_k |Y_k|^2
            int bad = 0;
#pragma unroll
            for (int q = 0; q < 16; ++q) {
                const float fd = out[t + 512 * q];
                bad |= (int)!(__builtin_fabsf(m2[q] - fd) <= kTd16Agree * __builtin_fmaxf(__builtin_fmaxf(m2[q], fd), l2)); // (a NaN anywhere: not agreed)
            }
            if (__syncthreads_or(bad)) { if (t == 0) a.fflags[fcur] = 2; ++n_esc; continue; } // (uniform)
#pragma unroll
            for (int q = 0; q < 16; ++q) out[t + 512 * q] = m2[q];
            GR4_T16_STAMP(7, fcur);
        }
    }
    // how many frames went on to the float64 evaluation: a running total in the page-locked word behind the launch's measurement (the host decides with it whether the stream is
    // better off on the time-domain kernel pair: chain.hip)
    if (t == 0 && n_esc && a.pw_host != nullptr) atomicAdd_system(reinterpret_cast<unsigned*>(a.pw_host) + 4, n_esc);
}
template <int MODE, int LOG2NF = 13>
__global__ __launch_bounds__(kT, 1) void chain_td16_kernel(ChainFdArgs a) { chain_td16_body<MODE, LOG2NF>(a); }

template <int MODE, int LOG2NF = 13>
__global__ __launch_bounds__(kT, 1) void chain_redo_kernel(ChainFdArgs a, int ntaps) { chain_redo_body<MODE, LOG2NF, false>(a, ntaps, nullptr); }
__global__ __launch_bounds__(kT, 1) void chain_redo_fold_kernel(ChainFdArgs a, ChainFdMulti m, int ntaps) { chain_redo_body<kModeMag2, 13, true>(a, ntaps, &m); }
__global__ __launch_bounds__(kT, 2) void chain_fd_multi_kernel(ChainFdArgs a, ChainFdMulti m) { chain_fd_body<kModeMag2, 13, true>(a, &m); }


int chain_fused_reset(struct ChainFused* c);
bool fir_f16_make_afrag(const float* taps, size_t ntaps, int* KS_out, std::vector<unsigned short>* af, size_t nch, int force_ks); // fir_f16.hip
// chain16.hip: the 16-wave kernel for the rectangular-window modes (|FFT(fir(x))|^2 and |FFT(x)|^2 at 8192 points)
struct Chain16;
int  chain16_create(Chain16** out, const float* H, const float* taps256);
void chain16_destroy(Chain16* c);
int  chain16_run(Chain16* c, const float* d_in, const float* d_hist256, size_t n_frames, float* d_out, unsigned max_wg, bool fft_only, hipStream_t st);
#ifdef GR4_FD_TIMING
static unsigned long long* g_dbg = nullptr;
#endif
struct ChainFused {
    size_t       ntaps = 0;
    DeviceBuffer d_H, d_twB, d_twC, d_taps, d_hist, d_win, d_twS, d_efrag;
    bool         windowed = false;
    int          small_log2n = 0; // 8..12: fftSize = 2^small_log2n < 8192, the launch unit stays an 8192-sample block
    DeviceBuffer d_stage_in, d_stage_out; // one zero-padded block for the tail of a span that is not a multiple of 8192 samples
    unsigned     max_wg   = 0; // 0 = one workgroup on every CU
    Chain16*     c16      = nullptr; // 8192-point plans: tables of the 16-wave kernel
    // dynamic-range guard (chain.hip / fir.hip decide with it): sampled input / output power of the last measured launch
    bool         measure  = false;
    DeviceBuffer d_pw;                 // {sum |x|^2, sum out, workgroups done}: accumulated by the kernel, re-armed by its last workgroup
    float*       h_pw     = nullptr;   // page-locked, device-mapped {in, out, sequence number of the launch that wrote them}
    float*       d_hpw    = nullptr;   // device view of h_pw
    unsigned     pw_seq   = 0;         // measured launches so far
    unsigned     pw_read  = 0;         // launches accounted for by the measurements handed out
    unsigned long long pw_word = 0;    // the last {in, out} pair seen
    unsigned     pw_seen  = 0;         // sequence number of the last measurement handed out
    hipStream_t  pw_stream = nullptr;  // stream of the last measured launch
    float        win_gain = 1.f;       // mean w[n]^2 of the window the measured output carries (1: none)
    float        win_mean = 1.f;       // mean w[n] (what a line keeps of its height)
    bool         redo     = false;     // measured launches also mark their frames one by one and chain_redo_kernel follows them (chain.hip, GR4HIP_GUARD_STRICT)
    bool         zero_hist = true;     // reset asked for (or nothing has run yet): the carried history is zeroed on the stream of the next call that reads it (common.hpp, the stream rule)
    unsigned     pw_floor = 0;         // measurements of launches up to this one belong to the stream before the last reset
    size_t       pw_items = 0;         // frames (x channels of a folded launch) the last measured launch judged
    size_t       judged = 0;           // frames judged since create / reset ...
    unsigned     f64_floor = 0;        // the running total of word 4 (frames that went on to float64) at the last reset
    DeviceBuffer d_fflags;             // one byte per frame of the last launch
    DeviceBuffer d_hfrag;              // the two-term f16 tap table of chain_td16_kernel (null: taps that form cannot carry -- the float64 evaluation takes every marked frame)
    bool         td16 = false;
    ~ChainFused() {
        if (c16) chain16_destroy(c16);
        if (h_pw) hip_quiet(hipHostFree(h_pw));
    }
};

int chain_fused_supported(size_t ntaps, size_t fft_size, int window, int algo) {
    if (algo != GR4HIP_CHAIN_FUSED_FD) return 0;
    const bool size_ok = fft_size == (size_t)kN || (is_pow2(fft_size) && fft_size >= 256 && fft_size <= 4096);
    return size_ok && ntaps >= 1 && ntaps <= 256 && window >= GR4HIP_WIN_NONE && window <= GR4HIP_WIN_KAISER;
}

template <typename T>
static int upload(DeviceBuffer& b, const std::vector<T>& h) {
    int rc = b.ensure(h.size() * sizeof(T));
    if (rc) return rc;
    GR4_HIP_TRY(upload_fresh(b.ptr, h.data(), h.size() * sizeof(T)));
    return GR4HIP_OK;
}

int chain_fused_create(ChainFused** out, const float* taps, size_t ntaps, size_t fft_size, int window, int algo) {
    if (!chain_fused_supported(ntaps, fft_size, window, algo)) { set_error("fused chain: unsupported configuration"); return GR4HIP_UNSUPPORTED; }
    auto* c = new (std::nothrow) ChainFused();
    GR4_REQUIRE(c, "out of host memory");
    c->ntaps = ntaps;
    std::vector<float> hp(256, 0.f);
    for (size_t k = 0; k < ntaps; ++k) hp[k] = taps[k];
    std::vector<float> H(2 * kN), twB(2 * 16 * 32), twC(2 * 16 * 512);
    for (int k = 0; k < kN; ++k) { // H[k] = sum_j b[j] e^{-2 pi i j k / N}, float64 accumulation, exact angle reduction
        double re = 0, im = 0;
        for (size_t j = 0; j < ntaps; ++j) {
            const double ang = -2.0 * M_PI * (double)((j * (size_t)k) % kN) / kN;
            re += hp[j] * std::cos(ang);
            im += hp[j] * std::sin(ang);
        }
        H[2 * k] = (float)re;
        H[2 * k + 1] = (float)im;
    }
    for (int r = 0; r < 16; ++r)
        for (int k = 0; k < 32; ++k) {
            const double ang = -2.0 * M_PI * (double)(r * k) / 512.0;
            twB[2 * (r * 32 + k)] = (float)std::cos(ang);
            twB[2 * (r * 32 + k) + 1] = (float)std::sin(ang);
        }
    for (int r = 0; r < 16; ++r)
        for (int i = 0; i < 512; ++i) {
            const double ang = -2.0 * M_PI * (double)(r * i) / 8192.0;
            twC[2 * (r * 512 + i)] = (float)std::cos(ang);
            twC[2 * (r * 512 + i) + 1] = (float)std::sin(ang);
        }
    int rc = upload(c->d_H, H);
    if (!rc) rc = upload(c->d_twB, twB);
    if (!rc) rc = upload(c->d_twC, twC);
    if (!rc) rc = upload(c->d_taps, hp);
    if (!rc) { // tap fragments of the correction e on the bf16 matrix pipe: [4 K quarters][2 K-steps][3 planes][64 lanes][8], A[j][u] = b_p[256 + j - u], b = h + m + l in bf16
        auto rne = [](float f) { unsigned u; std::memcpy(&u, &f, 4); u += 0x7FFFu + ((u >> 16) & 1u); return (unsigned short)(u >> 16); };
        auto tof = [](unsigned short h) { const unsigned u = (unsigned)h << 16; float f; std::memcpy(&f, &u, 4); return f; };
        std::vector<unsigned short> pl[3];
        for (auto& v : pl) v.assign(256, 0);
        for (int k = 0; k < 256; ++k) {
            const unsigned short h = rne(hp[k]);
            const float          r1 = hp[k] - tof(h);
            const unsigned short m = rne(r1);
            pl[0][k] = h; pl[1][k] = m; pl[2][k] = rne(r1 - tof(m));
        }
        std::vector<unsigned short> ef((size_t)4 * 2 * 3 * 64 * 8, 0);
        for (int kw = 0; kw < 4; ++kw)
            for (int s_ = 0; s_ < 2; ++s_)
                for (int p = 0; p < 3; ++p)
                    for (int l = 0; l < 64; ++l)
                        for (int t = 0; t < 8; ++t) {
                            const int k = 256 + (l & 15) - (64 * kw + 32 * s_ + 8 * (l >> 4) + t);
                            if (k >= 0 && k < 256) ef[((((size_t)kw * 2 + s_) * 3 + p) * 64 + l) * 8 + t] = pl[p][k];
                        }
        rc = upload(c->d_efrag, ef);
    }
    if (!rc) { // chain_td16_kernel's table (round 6): the same taps as two f16 terms, fragments of a 288-sample window
        std::vector<unsigned short> hf;
        int                         ks = 0;
        static const bool off = std::getenv("GR4HIP_CHAIN_NO_TD16") != nullptr; // developer switch: every marked frame on the float64 evaluation, round 5's behaviour
        if (!off && fir_f16_make_afrag(taps, ntaps, &ks, &hf, 1, kT16KS) && ks == kT16KS) { rc = upload(c->d_hfrag, hf); c->td16 = !rc; }
    }
    if (!rc && fft_size == (size_t)kN) rc = chain16_create(&c->c16, H.data(), hp.data());
    c->small_log2n = fft_size == (size_t)kN ? 0 : (int)ilog2(fft_size);
    c->windowed    = c->small_log2n != 0 || (window != GR4HIP_WIN_NONE && window != GR4HIP_WIN_RECTANGULAR);
    if (!rc && c->windowed) { // window[n mod fftSize] / 8192 over the whole block: the 1/N of the inverse 8192-point transform is exact
        std::vector<float> w(fft_size, 1.f), wt(kN);
        if (window != GR4HIP_WIN_NONE && window != GR4HIP_WIN_RECTANGULAR) rc = make_window(window, w.data(), fft_size, 1.6f); // fft.hpp:141: default beta
        for (int n = 0; n < kN; ++n) wt[n] = w[n % fft_size] * (1.0f / (float)kN);
        double g = 0, g1 = 0;
        for (size_t n = 0; n < fft_size; ++n) { g += (double)w[n] * w[n]; g1 += (double)w[n]; }
        c->win_gain = (float)(g / (double)fft_size);
        c->win_mean = (float)(g1 / (double)fft_size);
        if (!rc) rc = upload(c->d_win, wt);
    }
    if (!rc && c->small_log2n) {
        std::vector<float> ts(2 * fft_size);
        for (size_t k = 0; k < fft_size; ++k) {
            const double ang = -2.0 * M_PI * (double)k / (double)fft_size;
            ts[2 * k] = (float)std::cos(ang);
            ts[2 * k + 1] = (float)std::sin(ang);
        }
        rc = upload(c->d_twS, ts);
        if (!rc) rc = c->d_stage_in.ensure(kN * sizeof(float2));
        if (!rc) rc = c->d_stage_out.ensure(kN * sizeof(float));
    }
    if (!rc) rc = c->d_hist.ensure(256 * sizeof(float2));
    if (!rc) rc = chain_fused_reset(c);
    if (rc) { delete c; return rc; }
    *out = c;
    return GR4HIP_OK;
}

// Block::reset() runs on the block's worker between two work() calls (Block.hpp:606, 1296): here it is a host-side note, and the history is zeroed ON THE STREAM of
// the next call that reads it -- behind whatever that stream still has in flight for this handle (the carry of the launch before, chain_fused_run)
int chain_fused_reset(ChainFused* c) {
    c->zero_hist = true;
    c->pw_floor  = c->pw_seq; // what launches of the old stream measured decides nothing for the new one
    c->judged = 0;
    if (c->h_pw) c->f64_floor = reinterpret_cast<volatile unsigned*>(c->h_pw)[4];
    return GR4HIP_OK;
}
// the carried history as the next call must see it, in stream order
static int history_on(ChainFused* c, hipStream_t st) {
    if (c->zero_hist) {
        GR4_HIP_TRY(hipMemsetAsync(c->d_hist.ptr, 0, 256 * sizeof(float2), st));
        c->zero_hist = false;
    }
    return GR4HIP_OK;
}

// dynamic-range guard: this launch of `c` is a measured one (accumulators and the mapped result word exist from the first time on)
static int arm_measure(ChainFused* c, hipStream_t st) {
    if (!c->h_pw) {
        constexpr size_t words = kPwFrameSlots + 6 * kPwMaxWorkgroups; // 16 {in, out} slots, the done counter, the flag word, four verdict words per workgroup
        int rc = c->d_pw.ensure(words * sizeof(float));
        if (rc) return rc;
        GR4_HIP_TRY(hipMemsetAsync(c->d_pw.ptr, 0, words * sizeof(float), st)); // (in front of the first measured launch, on its stream)
        GR4_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&c->h_pw), 8 * sizeof(float), hipHostMallocMapped)); // {in, out, sequence number, frames marked, frames left to float64 (running total), -, -, -}
        std::memset(c->h_pw, 0, 8 * sizeof(float));
        GR4_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&c->d_hpw), c->h_pw, 0));
    }
    ++c->pw_seq;
    c->pw_stream = st;
    return GR4HIP_OK;
}

// the launches that follow a launch which has marked frames in a.fflags (same stream): chain_td16_kernel over the frames marked 1 (22-bit products on the f16 matrix pipe;
// it leaves a 2 on the frames that are beyond it), then chain_redo_kernel (float64 products) over what is left -- every marked frame when the taps have no f16 table
static int second_evaluations(ChainFused* c, ChainFdArgs a, size_t n_frames, hipStream_t st) {
    static PerDevice per_device;
    bool             first = false;
    int              dev = -1, n = per_device.current(&first, &dev);
    GR4_REQUIRE(n != 0, "fused chain: cannot query the current device");
    const int n_cu = n < 0 ? -n : n;
    if (first) {
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_td16_kernel<kModeMag2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kT16LdsBytes));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_td16_kernel<kModeWinMag2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kT16LdsBytes));
        per_device.done(dev, -n);
    }
    const unsigned rg = (unsigned)std::min<size_t>(n_frames, (size_t)n_cu);
    const int      nt = (int)c->ntaps;
    if (c->td16 && c->small_log2n == 0) { // (frames of fewer than 8192 points: float64 for every marked block -- the agreement test below is per 8192-point spectrum)
        a.hfrag    = static_cast<const unsigned short*>(c->d_hfrag.ptr);
        a.td16_thr = kTd16GuardRatio;
        if (c->windowed) hipLaunchKernelGGL(chain_td16_kernel<kModeWinMag2>, dim3(rg), dim3(kT), kT16LdsBytes, st, a);
        else hipLaunchKernelGGL(chain_td16_kernel<kModeMag2>, dim3(rg), dim3(kT), kT16LdsBytes, st, a);
        GR4_LAUNCH_CHECK();
        a.redo_min_flag = 2;
    }
    switch (c->small_log2n) {
    case 8: hipLaunchKernelGGL((chain_redo_kernel<kModeWinSmall, 8>), dim3(rg), dim3(kT), kRdLdsBytes, st, a, nt); break;
    case 9: hipLaunchKernelGGL((chain_redo_kernel<kModeWinSmall, 9>), dim3(rg), dim3(kT), kRdLdsBytes, st, a, nt); break;
    case 10: hipLaunchKernelGGL((chain_redo_kernel<kModeWinSmall, 10>), dim3(rg), dim3(kT), kRdLdsBytes, st, a, nt); break;
    case 11: hipLaunchKernelGGL((chain_redo_kernel<kModeWinSmall, 11>), dim3(rg), dim3(kT), kRdLdsBytes, st, a, nt); break;
    case 12: hipLaunchKernelGGL((chain_redo_kernel<kModeWinSmall, 12>), dim3(rg), dim3(kT), kRdLdsBytes, st, a, nt); break;
    default:
        if (c->windowed) hipLaunchKernelGGL(chain_redo_kernel<kModeWinMag2>, dim3(rg), dim3(kT), kRdLdsBytes, st, a, nt);
        else hipLaunchKernelGGL(chain_redo_kernel<kModeMag2>, dim3(rg), dim3(kT), kRdLdsBytes, st, a, nt);
    }
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

// hist256 == nullptr: the chain's own carried history (updated after the launch); otherwise 256 complex samples preceding d_in, and
// the output is the filtered stream itself (complex) instead of |FFT|^2
static int chain_fused_run(ChainFused* c, const float* d_in, const float* hist256, size_t n_frames, float* d_out, hipStream_t st, bool fir_mode, bool carry_hist,
                           bool fft_only = false, const float* fft_window = nullptr, bool fft_spectrum = false) {
    if (!hist256) { if (const int rc = history_on(c, st)) return rc; }
    {
        const int use16 = dev_switch(kDevChain16);
        const bool plain_chain = !fir_mode && !fft_only && !c->windowed && c->small_log2n == 0;
        const bool plain_fft   = fft_only && !fft_window && !fft_spectrum;
        if (use16 && c->c16 && (plain_chain || plain_fft) && !(c->measure && !fft_only)) { // (the dynamic-range guard samples its powers in the 8-wave kernel)
            const float* hist = hist256 ? hist256 : static_cast<const float*>(c->d_hist.ptr);
            int rc = chain16_run(c->c16, d_in, hist, n_frames, d_out, c->max_wg, plain_fft, st);
            if (rc) return rc;
            if (carry_hist) GR4_HIP_TRY(hipMemcpyAsync(c->d_hist.ptr, reinterpret_cast<const float2*>(d_in) + n_frames * (size_t)kN - 256, 256 * sizeof(float2), hipMemcpyDeviceToDevice, st));
            return GR4HIP_OK;
        }
    }
    ChainFdArgs a{};
    a.x        = reinterpret_cast<const float2*>(d_in);
    a.hist     = hist256 ? reinterpret_cast<const float2*>(hist256) : static_cast<const float2*>(c->d_hist.ptr);
    a.H        = static_cast<const float2*>(c->d_H.ptr);
    a.twB      = static_cast<const float2*>(c->d_twB.ptr);
    a.twC      = static_cast<const float2*>(c->d_twC.ptr);
    a.taps     = static_cast<const float*>(c->d_taps.ptr);
    a.efrag    = c->d_efrag.ptr;
    a.win      = fft_only ? fft_window : static_cast<const float*>(c->d_win.ptr);
    a.twS      = static_cast<const float2*>(c->d_twS.ptr);
    a.out      = d_out;
    a.n_frames = (long)n_frames;
    a.dbg      = nullptr;
    a.pw       = nullptr;
    const bool measure = c->measure && !fft_only;
    if (measure) {
        int rc = arm_measure(c, st);
        if (rc) return rc;
        c->pw_items = n_frames;
        c->judged += n_frames;
        a.pw      = static_cast<float*>(c->d_pw.ptr);
        a.pw_host = c->d_hpw;
        a.pw_seq  = c->pw_seq;
        a.pw_thr  = fir_mode ? kGuardFirFrameThreshold : kGuardFrameThreshold * (float)(c->small_log2n ? (1 << c->small_log2n) : kN) * c->win_gain; // (the scale chain_fused_power_ratio takes out)
        {   // R4 = scale S_in / (sqrt(8192) sqrt(S4)), scale = nf w2 (sums over one 8192-sample block); marked when R4 / kGuardR4Max + T' / kGuardPeakMax > 1
            const double scale = (double)(c->small_log2n ? (1 << c->small_log2n) : kN) * (double)c->win_gain;
            const double r4max = c->small_log2n ? kGuardR4MaxSmall : kGuardR4Max, tmax = c->small_log2n ? kGuardPeakMaxSmall : kGuardPeakMax;
            a.pw_c4 = (float)(scale / (std::sqrt((double)kN) * r4max)); // R4 / R4max = pw_c4 S_in / sqrt(S4)
            // T' = 2 wg^2 (peak nf / 8192)^2 / sqrt(S4 / 8192) > Tmax  <=>  S4 < peak^4 x 4 wg^4 nf^4 / (Tmax^2 8192^3)   (peak: of the 8192-point X of the block)
            const double nfr = (double)(c->small_log2n ? (1 << c->small_log2n) : kN) / (double)kN, wg2 = (double)c->win_mean * c->win_mean;
            a.pw_c5i = (float)(2.0 * wg2 * nfr * nfr * std::sqrt((double)kN) / tmax); // T' / Tmax = pw_c5i peak^2 / sqrt(S4)
            a.no_tier = (c->td16 && c->small_log2n == 0) ? 0 : 1; // (second_evaluations: chain_td16_kernel first, where the taps have an f16 table and the frames 8192 points)
        }
        if (c->redo && !fir_mode) {
            rc = c->d_fflags.ensure(n_frames);
            if (rc) return rc;
            a.fflags = static_cast<unsigned char*>(c->d_fflags.ptr);
        }
    }
#ifdef GR4_FD_TIMING
    if (!g_dbg) GR4_HIP_TRY(hipMalloc(&g_dbg, (size_t)1 << 26));
    if (n_frames * 8 * 16 * 8 <= ((size_t)1 << 26)) a.dbg = g_dbg;
#endif
    // two frame buffers, two tails, e | partial tiles (+ WIN: pass-B twiddle table), planar padded d, taps
    constexpr size_t lds_base = (size_t)(2 * kSLen + 512 + 256) * sizeof(float2) + (size_t)(4 * 2 * 256 + 2 * kDPad + 272) * sizeof(float);
    constexpr size_t lds_win  = lds_base + ((GR4_E_BF16 && GR4_E_BF16_WIN) ? 960 * sizeof(float) + 6 * 512 * sizeof(unsigned short) - (2 * kDPad + 272) * sizeof(float) : 1024 * sizeof(float)); // = 160 KiB exactly with the bf16 planes
    constexpr size_t lds_ebf  = lds_base + (GR4_E_BF16 ? 6 * 512 * sizeof(unsigned short) - (2 * kDPad + 272) * sizeof(float) : 0) + kGvBytes; // non-windowed filter modes: six bf16 planes of Dz instead of Dre / Dim / hl, + the verdict words
    static_assert(!GR4_E_BF16 || lds_ebf == (size_t)kLdsEbfBytes + kGvBytes, "the kernel's verdict words sit behind the image");
    static_assert(lds_win <= 160 * 1024 && lds_ebf <= 160 * 1024, "LDS budget of one CU");
    const size_t lds  = (c->windowed && !fir_mode && !fft_only) ? lds_win : (fft_only ? lds_base : lds_ebf);
    static PerDevice per_device; // LDS opt-in and CU count, once per device this process uses
    bool             first = false;
    int              dev = -1, n_cu = per_device.current(&first, &dev);
    GR4_REQUIRE(n_cu != 0, "fused chain: cannot query the current device");
    if (first) {
        n_cu = -n_cu;
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fd_kernel<kModeMag2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ebf));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fd_kernel<kModeWinMag2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_win));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fd_kernel<kModeFir>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_ebf));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fd_kernel<kModeFftMag2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_base));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fd_kernel<kModeFftWinMag2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_base));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fd_kernel<kModeFftSpec>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_base));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fd_kernel<kModeFftWinSpec>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_base));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fd_kernel<kModeWinSmall, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_win));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fd_kernel<kModeWinSmall, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_win));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fd_kernel<kModeWinSmall, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_win));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fd_kernel<kModeWinSmall, 11>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_win));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fd_kernel<kModeWinSmall, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_win));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_redo_kernel<kModeMag2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRdLdsBytes));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_redo_kernel<kModeWinMag2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRdLdsBytes));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_redo_kernel<kModeWinSmall, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRdLdsBytes));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_redo_kernel<kModeWinSmall, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRdLdsBytes));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_redo_kernel<kModeWinSmall, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRdLdsBytes));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_redo_kernel<kModeWinSmall, 11>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRdLdsBytes));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_redo_kernel<kModeWinSmall, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRdLdsBytes));
        per_device.done(dev, n_cu);
    }
    const size_t   wgs  = c->max_wg ? std::min<size_t>(c->max_wg, (size_t)n_cu) : (size_t)n_cu;
    const unsigned grid = (unsigned)std::min<size_t>(n_frames, wgs); // one resident workgroup per CU (or fewer: gr4hip_chain_set_max_workgroups)
    if (fft_only && fft_spectrum && fft_window) hipLaunchKernelGGL(chain_fd_kernel<kModeFftWinSpec>, dim3(grid), dim3(kT), lds, st, a);
    else if (fft_only && fft_spectrum) hipLaunchKernelGGL(chain_fd_kernel<kModeFftSpec>, dim3(grid), dim3(kT), lds, st, a);
    else if (fft_only && fft_window) hipLaunchKernelGGL(chain_fd_kernel<kModeFftWinMag2>, dim3(grid), dim3(kT), lds, st, a);
    else if (fft_only) hipLaunchKernelGGL(chain_fd_kernel<kModeFftMag2>, dim3(grid), dim3(kT), lds, st, a);
    else if (fir_mode) hipLaunchKernelGGL(chain_fd_kernel<kModeFir>, dim3(grid), dim3(kT), lds, st, a);
    else if (c->small_log2n == 8) hipLaunchKernelGGL((chain_fd_kernel<kModeWinSmall, 8>), dim3(grid), dim3(kT), lds, st, a);
    else if (c->small_log2n == 9) hipLaunchKernelGGL((chain_fd_kernel<kModeWinSmall, 9>), dim3(grid), dim3(kT), lds, st, a);
    else if (c->small_log2n == 10) hipLaunchKernelGGL((chain_fd_kernel<kModeWinSmall, 10>), dim3(grid), dim3(kT), lds, st, a);
    else if (c->small_log2n == 11) hipLaunchKernelGGL((chain_fd_kernel<kModeWinSmall, 11>), dim3(grid), dim3(kT), lds, st, a);
    else if (c->small_log2n == 12) hipLaunchKernelGGL((chain_fd_kernel<kModeWinSmall, 12>), dim3(grid), dim3(kT), lds, st, a);
    else if (c->windowed) hipLaunchKernelGGL(chain_fd_kernel<kModeWinMag2>, dim3(grid), dim3(kT), lds, st, a);
    else hipLaunchKernelGGL(chain_fd_kernel<kModeMag2>, dim3(grid), dim3(kT), lds, st, a);
    GR4_LAUNCH_CHECK();
    if (a.fflags != nullptr) { // the marked frames again in the time domain, behind the launch that marked them
        if (const int rc = second_evaluations(c, a, n_frames, st)) return rc;
    }
    // carry the last 256 input samples for the next call's first frame (stream-ordered after the kernel's reads)
    if (carry_hist) GR4_HIP_TRY(hipMemcpyAsync(c->d_hist.ptr, a.x + n_frames * (size_t)kN - 256, 256 * sizeof(float2), hipMemcpyDeviceToDevice, st));
    return GR4HIP_OK;
}

// n_frames counts FFT frames of the plan's fftSize.  At fftSize < 8192 whole 8192-sample blocks go through the kernel directly; the frames
// behind the last whole block (< 8192 samples) are copied into a zero-padded staging block, transformed, and only their spectra copied out.
int chain_fused_process(ChainFused* c, const float* d_in, size_t n_frames, float* d_mag2, hipStream_t st) {
    const auto run = [&](const float* in, size_t blocks, float* out, bool carry) { return chain_fused_run(c, in, nullptr, blocks, out, st, false, carry); };
    if (c->small_log2n == 0) return run(d_in, n_frames, d_mag2, true);
    const size_t nf = (size_t)1 << c->small_log2n, per_block = kN / nf;
    const size_t blocks = n_frames / per_block, rem = n_frames % per_block; // rem fft-frames = rem * nf samples (>= 256 each)
    if (blocks) {
        int rc = run(d_in, blocks, d_mag2, true);
        if (rc) return rc;
    }
    if (rem) {
        const float* tail = d_in + blocks * (size_t)kN * 2;
        GR4_HIP_TRY(hipMemsetAsync(c->d_stage_in.ptr, 0, kN * sizeof(float2), st));
        GR4_HIP_TRY(hipMemcpyAsync(c->d_stage_in.ptr, tail, rem * nf * sizeof(float2), hipMemcpyDeviceToDevice, st));
        int rc = run(static_cast<const float*>(c->d_stage_in.ptr), 1, static_cast<float*>(c->d_stage_out.ptr), false);
        if (rc) return rc;
        GR4_HIP_TRY(hipMemcpyAsync(d_mag2 + blocks * (size_t)kN, c->d_stage_out.ptr, rem * nf * sizeof(float), hipMemcpyDeviceToDevice, st));
        // history for the next call: the last 256 REAL input samples (rem * nf >= 256)
        GR4_HIP_TRY(hipMemcpyAsync(c->d_hist.ptr, tail + (rem * nf - 256) * 2, 256 * sizeof(float2), hipMemcpyDeviceToDevice, st));
    }
    return GR4HIP_OK;
}
// |FFT_8192(window x frame)|^2 for n_frames frames, no filter: the FFT block's mag2 output on the frame pipeline of this kernel (d_window: 8192
// floats or null for None / Rectangular)
int chain_fused_fft_mag2(ChainFused* c, const float* d_in, size_t n_frames, float* d_mag2, const float* d_window, hipStream_t st) {
    GR4_REQUIRE(c->small_log2n == 0, "chain_fused_fft_mag2: needs an 8192-point plan");
    return chain_fused_run(c, d_in, nullptr, n_frames, d_mag2, st, false, false, true, d_window);
}
// FFT_8192(window x frame) itself (interleaved re, im) on the same pipeline
int chain_fused_fft_spectrum(ChainFused* c, const float* d_in, size_t n_frames, float* d_spectrum, const float* d_window, hipStream_t st) {
    GR4_REQUIRE(c->small_log2n == 0, "chain_fused_fft_spectrum: needs an 8192-point plan");
    return chain_fused_run(c, d_in, nullptr, n_frames, d_spectrum, st, false, false, true, d_window, true);
}
int chain_fused_fir(ChainFused* c, const float* d_in, const float* d_hist256, size_t n_frames, float* d_y, hipStream_t st) {
    GR4_REQUIRE(d_hist256 && c->small_log2n == 0, "chain_fused_fir: needs a history and an 8192-point plan");
    return chain_fused_run(c, d_in, d_hist256, n_frames, d_y, st, true, false);
}

// n fused chains (8192-point plans, rectangular window) in ONE launch: n_frames frames of every d_in[i].
//   d_sum != nullptr (needs shared_taps): only the combiner output sum_i |FFT(fir(x_i))|^2 is written (math::Add's left fold, kept in registers);
//   otherwise d_out[i] receives chain i's spectra and workgroup b works for chain b mod n.
// The guard's powers: with the fold, of all channels together into chain 0's slots (the ratio that matters for the delivered sum); otherwise per chain.
int chain_fused_redo(ChainFused* c, const float* d_in, const float* d_hist, int hist_len, size_t n_samples, float* d_out, const unsigned char* d_flags, int flags_per_block, hipStream_t st);
// redo: the measured chains' frames are marked one by one and evaluated again in the time domain behind the launch (chain_redo_kernel per chain; the fold: chain_redo_fold_kernel)
int chain_fused_process_multi(ChainFused* const* cs, size_t n, bool shared_taps, const float* const* d_in, size_t n_frames, float* const* d_out, float* d_sum, hipStream_t st, bool redo) {
    GR4_REQUIRE(n >= 1 && n <= (size_t)kMaxMulti, "chain multi: 1 .. %d chains per launch", kMaxMulti);
    const bool fold = d_sum != nullptr;
    GR4_REQUIRE(!fold || shared_taps, "chain multi: the in-register fold needs identical taps on every chain");
    GR4_REQUIRE(fold || d_out, "chain multi: no output");
    for (size_t i = 0; i < n; ++i) GR4_REQUIRE(cs[i] && cs[i]->small_log2n == 0 && !cs[i]->windowed, "chain multi: 8192-point rectangular-window fused chains only");
    ChainFdArgs  a{};
    ChainFdMulti m{};
    ChainFused*  c0 = cs[0];
    for (size_t i = 0; i < n; ++i) { if (const int rc = history_on(cs[i], st)) return rc; }
    a.x = reinterpret_cast<const float2*>(d_in[0]);
    a.hist = static_cast<const float2*>(c0->d_hist.ptr);
    a.H = static_cast<const float2*>(c0->d_H.ptr);
    a.twB = static_cast<const float2*>(c0->d_twB.ptr);
    a.twC = static_cast<const float2*>(c0->d_twC.ptr);
    a.taps = static_cast<const float*>(c0->d_taps.ptr);
    a.efrag = c0->d_efrag.ptr;
    a.out = fold ? d_sum : d_out[0];
    a.n_frames = (long)n_frames;
    m.n_ch = (int)n;
    m.fold_ch = fold ? (int)n : 1;
    for (size_t i = 0; i < n; ++i) {
        ChainFused* c = cs[i];
        m.xs[i] = reinterpret_cast<const float2*>(d_in[i]);
        m.hists[i] = static_cast<const float2*>(c->d_hist.ptr);
        m.Hs[i] = static_cast<const float2*>(c->d_H.ptr);
        m.efrags[i] = c->d_efrag.ptr;
        m.outs[i] = fold ? d_sum : d_out[i];
        const bool measure = c->measure && (!fold || i == 0);
        if (measure) {
            int rc = arm_measure(c, st);
            if (rc) return rc;
            c->pw_items = fold ? n_frames * n : n_frames;
            c->judged += c->pw_items;
            m.pws[i] = static_cast<float*>(c->d_pw.ptr);
            m.pw_hosts[i] = c->d_hpw;
            m.pw_seqs[i] = c->pw_seq;
            if (redo) { // (set-only in the kernel: zeroed here)
                rc = c->d_fflags.ensure(n_frames);
                if (rc) return rc;
                GR4_HIP_TRY(hipMemsetAsync(c->d_fflags.ptr, 0, n_frames, st));
                m.fflags[i] = static_cast<unsigned char*>(c->d_fflags.ptr);
            }
        }
    }
    if (fold) { a.pw = m.pws[0]; a.pw_host = m.pw_hosts[0]; a.pw_seq = m.pw_seqs[0]; a.fflags = m.fflags[0]; }
    a.pw_thr = kGuardFrameThreshold * (float)kN; // (8192-point rectangular-window chains only: window gain 1)
    a.pw_c4  = std::sqrt((float)kN) / kGuardR4Max;
    a.pw_c5i = 2.0f * std::sqrt((float)kN) / kGuardPeakMax;
    a.no_tier = 1; // (the multi launch's marked frames go to chain_redo_kernel / chain_redo_fold_kernel)
    constexpr size_t lds = (size_t)kLdsEbfBytes + kGvBytes; // = lds_ebf of chain_fused_run
    static PerDevice per_device;
    bool             first = false;
    int              dev = -1, n_cu = per_device.current(&first, &dev);
    GR4_REQUIRE(n_cu != 0, "chain multi: cannot query the current device");
    if (first) {
        n_cu = -n_cu;
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fd_multi_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        per_device.done(dev, n_cu);
    }
    size_t wgs = c0->max_wg ? std::min<size_t>(c0->max_wg, (size_t)n_cu) : (size_t)n_cu;
    unsigned grid;
    if (fold) {
        grid = (unsigned)std::min<size_t>(n_frames, wgs);
    } else { // the same number of workgroups for every chain
        const size_t per = std::min<size_t>(std::max<size_t>(wgs / n, 1), n_frames);
        grid = (unsigned)(per * n);
    }
    hipLaunchKernelGGL(chain_fd_multi_kernel, dim3(grid), dim3(kT), lds, st, a, m);
    GR4_LAUNCH_CHECK();
    if (redo) { // the marked frames again in the time domain, from the histories the call started with (the carry below is stream-ordered behind these launches)
        if (fold) {
            if (a.fflags != nullptr) {
                if (first) GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_redo_fold_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRdLdsBytes));
                hipLaunchKernelGGL(chain_redo_fold_kernel, dim3((unsigned)std::min<size_t>(n_frames, (size_t)n_cu)), dim3(kT), kRdLdsBytes, st, a, m, (int)c0->ntaps);
                GR4_LAUNCH_CHECK();
            }
        } else {
            for (size_t i = 0; i < n; ++i)
                if (m.fflags[i] != nullptr) {
                    const int rc = chain_fused_redo(cs[i], d_in[i], static_cast<const float*>(cs[i]->d_hist.ptr), 256, n_frames * (size_t)kN, d_out[i], m.fflags[i], 1, st);
                    if (rc) return rc;
                }
        }
    }
    for (size_t i = 0; i < n; ++i) // carry every chain's last 256 input samples (stream-ordered after the kernel's reads)
        GR4_HIP_TRY(hipMemcpyAsync(cs[i]->d_hist.ptr, m.xs[i] + n_frames * (size_t)kN - 256, 256 * sizeof(float2), hipMemcpyDeviceToDevice, st));
    return GR4HIP_OK;
}

bool chain_fused_multi_capable(const ChainFused* c) { return c->small_log2n == 0 && !c->windowed; }
void chain_fused_destroy(ChainFused* c) { delete c; }
// dynamic-range guard: sampled power ratio (output / input, window gain taken out) of the most recent measured launch.
// wait: synchronise on that launch; otherwise only report it when it has already finished.  Returns 1 with *ratio set, 0 if nothing (new) is available.
void chain_fused_set_measure(ChainFused* c, bool on) { c->measure = on; }
void chain_fused_set_redo(ChainFused* c, bool on) { c->redo = on; }

// (chain.hip) the second evaluation for ANOTHER kernel's launch -- the fused time-domain chain (chain_td.hip), which judges 4096-sample segments: the blocks of the span
// d_in[0 .. n_samples) that hold a marked segment again with float64 products, over d_out.  `c` supplies the tables (taps, window, twiddles) of the same chain
// configuration; d_hist: the hist_len samples in front of d_in.
int chain_fused_redo(ChainFused* c, const float* d_in, const float* d_hist, int hist_len, size_t n_samples, float* d_out, const unsigned char* d_flags, int flags_per_block, hipStream_t st) {
    if (n_samples == 0) return GR4HIP_OK;
    ChainFdArgs a{};
    a.x    = reinterpret_cast<const float2*>(d_in);
    a.hist = reinterpret_cast<const float2*>(d_hist);
    a.twB  = static_cast<const float2*>(c->d_twB.ptr);
    a.twC  = static_cast<const float2*>(c->d_twC.ptr);
    a.taps = static_cast<const float*>(c->d_taps.ptr);
    a.win  = static_cast<const float*>(c->d_win.ptr);
    a.twS  = static_cast<const float2*>(c->d_twS.ptr);
    a.out  = d_out;
    a.n_frames = (long)ceil_div(n_samples, (size_t)kN);
    a.fflags   = const_cast<unsigned char*>(d_flags);
    a.redo_hist_len = hist_len; a.redo_flags_per_block = flags_per_block; a.redo_n = (long)n_samples;
    static PerDevice per_device;
    bool             first = false;
    int              dev = -1, n_cu = per_device.current(&first, &dev);
    GR4_REQUIRE(n_cu != 0, "fused chain: cannot query the current device");
    if (first) {
        n_cu = -n_cu;
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_redo_kernel<kModeMag2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRdLdsBytes));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_redo_kernel<kModeWinMag2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRdLdsBytes));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_redo_kernel<kModeWinSmall, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRdLdsBytes));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_redo_kernel<kModeWinSmall, 9>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRdLdsBytes));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_redo_kernel<kModeWinSmall, 10>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRdLdsBytes));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_redo_kernel<kModeWinSmall, 11>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRdLdsBytes));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_redo_kernel<kModeWinSmall, 12>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kRdLdsBytes));
        per_device.done(dev, n_cu);
    }
    const unsigned rg = (unsigned)std::min<size_t>((size_t)a.n_frames, (size_t)n_cu);
    const int      nt = (int)c->ntaps;
    switch (c->small_log2n) {
    case 8: hipLaunchKernelGGL((chain_redo_kernel<kModeWinSmall, 8>), dim3(rg), dim3(kT), kRdLdsBytes, st, a, nt); break;
    case 9: hipLaunchKernelGGL((chain_redo_kernel<kModeWinSmall, 9>), dim3(rg), dim3(kT), kRdLdsBytes, st, a, nt); break;
    case 10: hipLaunchKernelGGL((chain_redo_kernel<kModeWinSmall, 10>), dim3(rg), dim3(kT), kRdLdsBytes, st, a, nt); break;
    case 11: hipLaunchKernelGGL((chain_redo_kernel<kModeWinSmall, 11>), dim3(rg), dim3(kT), kRdLdsBytes, st, a, nt); break;
    case 12: hipLaunchKernelGGL((chain_redo_kernel<kModeWinSmall, 12>), dim3(rg), dim3(kT), kRdLdsBytes, st, a, nt); break;
    default:
        if (c->windowed) hipLaunchKernelGGL(chain_redo_kernel<kModeWinMag2>, dim3(rg), dim3(kT), kRdLdsBytes, st, a, nt);
        else hipLaunchKernelGGL(chain_redo_kernel<kModeMag2>, dim3(rg), dim3(kT), kRdLdsBytes, st, a, nt);
    }
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}
int  chain_fused_power_ratio(ChainFused* c, bool wait, bool fir_output, float* ratio, float* marked_fraction, float* float64_fraction) {
    if (!c->h_pw || c->pw_seq == c->pw_read || c->pw_seq == c->pw_floor) return 0;
    volatile unsigned* seqw = reinterpret_cast<volatile unsigned*>(c->h_pw) + 2; // sequence number of the launch whose pair the word holds
    if (wait) { // spin on the mapped word the launch's last workgroup writes; every 4096 polls check that the stream has not simply failed / finished without it
        for (unsigned long spins = 1; *seqw != c->pw_seq; ++spins) {
            if ((spins & 4095) == 0 && hipStreamQuery(c->pw_stream) != hipErrorNotReady) {
                if (hipStreamSynchronize(c->pw_stream) != hipSuccess) return 0;
                if (*seqw != c->pw_seq) return 0; // (a launch that measured nothing)
                break;
            }
            __builtin_ia32_pause();
        }
    }
    const unsigned long long word = *reinterpret_cast<volatile unsigned long long*>(c->h_pw); // {in, out} of the most recent finished launch, stored whole
    if (!wait && (*seqw == c->pw_seen || (int)(*seqw - c->pw_floor) <= 0)) return 0;            // nothing new has arrived (or only from before the last reset)
    c->pw_seen = *seqw;
    c->pw_word = word;
    c->pw_read = c->pw_seq; // (with wait: exactly; without: at least one newer launch has reported)
    float pair[2];
    std::memcpy(pair, &word, sizeof(pair));
    const double in = pair[0], out = pair[1];
    // spectra: sum_k |Y_k|^2 = N sum_n |w_n y_n|^2 ~ N mean(w^2) sum |y|^2; the complex FIR output is y itself
    const double nfft = c->small_log2n ? (double)(1 << c->small_log2n) : (double)kN; // the spectra summed are fftSize-point ones
    *ratio = in > 0 ? (float)(out / (in * (fir_output ? 1.0 : nfft * c->win_gain))) : 1.f;
    // the launch-wide ratio can hide a frame: an interferer that arrives late in a long span barely moves the sums.  Every frame is judged by itself in the kernel
    // (a quarter of its points, per wave); a launch with ONE frame below the threshold reports below the threshold
    const unsigned n_marked = reinterpret_cast<volatile unsigned*>(c->h_pw)[3];
    if (float64_fraction) { // of the frames judged since the last reset, how many took the float64 evaluation (as far as the kernels' running totals have arrived)
        const size_t n64 = (size_t)(reinterpret_cast<volatile unsigned*>(c->h_pw)[4] - c->f64_floor); // (chain_td16_kernel's escalations; the marked frames of launches without that tier)
        *float64_fraction = c->judged ? std::min(1.f, (float)n64 / (float)c->judged) : 0.f;
    }
    if (marked_fraction) *marked_fraction = c->pw_items ? std::min(1.f, (float)n_marked / (float)c->pw_items) : (n_marked ? 1.f : 0.f);
    // kModeFir (y itself is the output): the launch reports below the threshold as soon as ONE frame is.  The |.|^2 modes (round 6) report the power ratio as it is: what their
    // frames are judged on is the fourth-moment statistic (kGuardR4Max), and what the caller decides on is the fraction of frames that were marked
    if (fir_output && n_marked != 0u && *ratio >= kGuardFirFrameThreshold) *ratio = 0.5f * kGuardFirFrameThreshold;
    return 1;
}
// the 256 samples before the next call's first frame, valid for work enqueued on `st` behind this call (null: the pending zeroing could not be enqueued)
const float* chain_fused_history(ChainFused* c, hipStream_t st) { return history_on(c, st) ? nullptr : static_cast<const float*>(c->d_hist.ptr); }
int chain_fused_set_history(ChainFused* c, const float* d_hist256, hipStream_t st) {
    c->zero_hist = false;
    GR4_HIP_TRY(hipMemcpyAsync(c->d_hist.ptr, d_hist256, 256 * sizeof(float2), hipMemcpyDeviceToDevice, st));
    return GR4HIP_OK;
}
void chain_fused_set_max_workgroups(ChainFused* c, unsigned n) { c->max_wg = n; }

#ifdef GR4_FD_TIMING
} // namespace gr4
extern "C" int gr4hip_dbg_fd_timing(unsigned long long* h_out, size_t n_frames) { // developer-only, not part of the ABI
    if (!gr4::g_dbg) return GR4HIP_ERROR;
    gr4::hip_quiet(hipDeviceSynchronize());
    return hipMemcpy(h_out, gr4::g_dbg, n_frames * 8 * 16 * 8, hipMemcpyDeviceToHost) == hipSuccess ? GR4HIP_OK : GR4HIP_RUNTIME_ERROR;
}
namespace gr4 {
#endif

} // namespace gr4
