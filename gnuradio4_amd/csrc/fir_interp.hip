// fir_interp.hip -- polyphase interpolating FIR for gfx950 (BASELINE.json north_star: "decimating / interpolating FIR").
//
// The reference has no interpolating filter block -- only the rate machinery a block would declare, Resampling<1, L>
// (core/include/gnuradio-4.0/annotated.hpp:121-128, chunk bookkeeping Block.hpp:1576-1636).  Definition: zero-stuff the
// input by L, run fir_filter's sum (blocks/filter/.../time_domain_filter.hpp:44-47) at the output rate, gain L:
//     u[n] = x[n / L] if n % L == 0 else 0,       y[n] = L sum_k b[k] u[n - k]
// Only every L-th product is non-zero, so output n = m L + p is branch p of a polyphase bank at the INPUT rate:
//     y[m L + p] = sum_q (L b[q L + p]) x[m - q],   q < Kp = ceil(K / L)
// i.e. L short FIRs over the same input window.  Two evaluations:
//  * spans of >= 32768 outputs, L in {2, 4, 8, 16} or any L except 3, 5, 6: a block-Toeplitz contraction on the f32 matrix pipe (fir_interp_mfma_kernel
//    below) -- write-bound at short branches (4.2 .. 5.4 TB/s), MFMA-bound at long ones (L = 8, K = 1024: 124 useful TFLOP/s);
//  * otherwise (short spans; L = 3, 5, 6; branches longer than 272 taps): fir_interp_kernel<L, R, S> -- a lane owns R consecutive input positions and
//    all L phases of them: R L accumulators, one sliding register window of the staged input (one LDS read per tap step), the L branch taps of step q
//    are wave-uniform (scalar loads), the R L outputs of a lane are contiguous in memory; FP32 FMA rate at long filters (2 K / L flop per output),
//    HBM at short ones ((4 + 4 L) B per real input sample).  Factors it is not instantiated for take one output per lane straight from the L2.
// History: the last Kp - 1 input samples (capacity max(32, bit_ceil(Kp)) like HistoryBuffer, :36-42).
#include "common.hpp"
#include "buffer_ops.hpp"

#include <algorithm>
#include <cstdlib>

namespace gr4 {

constexpr int kIpBS = 256; // lanes per workgroup

template <int S> struct ip_vec { using type = float; };
template <> struct ip_vec<2> { using type = float2; };
__device__ __forceinline__ float  ip_fma(float w, float x, float a) { return fmaf(w, x, a); }
__device__ __forceinline__ float2 ip_fma(float w, float2 x, float2 a) { return make_float2(fmaf(w, x.x, a.x), fmaf(w, x.y, a.y)); }

// taps: [Kp][L] (q-major), already times L.  hist: hcap samples, hist[h] = stream position -hcap + h.
template <int L, int R, int S>
__global__ __launch_bounds__(kIpBS) void fir_interp_kernel(const typename ip_vec<S>::type* __restrict__ x, const typename ip_vec<S>::type* __restrict__ hist, int hcap,
                                                           const float* __restrict__ taps, int Kp, typename ip_vec<S>::type* __restrict__ y, long n_in) {
    using T = typename ip_vec<S>::type;
    extern __shared__ __attribute__((aligned(16))) float ip_smem[];
    T*         xs   = reinterpret_cast<T*>(ip_smem); // [H + TM]: positions tile0 - H .. tile0 + TM - 1
    const int  H    = Kp - 1;
    const long TM   = (long)kIpBS * R;
    const long tile0 = (long)blockIdx.x * TM;
    for (long i = threadIdx.x; i < H + TM; i += kIpBS) {
        const long pos = tile0 - H + i;
        T          v   = T{};
        if (pos >= 0) { if (pos < n_in) v = x[pos]; }
        else if (pos >= -(long)hcap) v = hist[hcap + pos];
        xs[i] = v;
    }
    __syncthreads();
    const long m0 = tile0 + (long)threadIdx.x * R;
    if (m0 >= n_in) return;
    T acc[R][L];
#pragma unroll
    for (int r = 0; r < R; ++r)
#pragma unroll
        for (int p = 0; p < L; ++p) acc[r][p] = T{};
    T         win[R]; // win[r] = x[m0 + r - q]
    const T*  xp = xs + H + threadIdx.x * R;
#pragma unroll
    for (int r = 0; r < R; ++r) win[r] = xp[r];
    for (int q = 0; q < Kp; ++q) {
        const float* tq = taps + (long)q * L; // wave-uniform: scalar loads
#pragma unroll
        for (int p = 0; p < L; ++p) {
            const float w = tq[p];
#pragma unroll
            for (int r = 0; r < R; ++r) acc[r][p] = ip_fma(w, win[r], acc[r][p]);
        }
#pragma unroll
        for (int r = R - 1; r > 0; --r) win[r] = win[r - 1];
        if (q + 1 < Kp) win[0] = xp[-(q + 1)];
    }
    T* yo = y + m0 * L;
#pragma unroll
    for (int r = 0; r < R; ++r)
        if (m0 + r < n_in) {
#pragma unroll
            for (int p = 0; p < L; ++p) yo[r * L + p] = acc[r][p];
        }
}

// any interpolation factor: one output per lane straight from global memory (the L2 holds the window); a corner, not the hot path
template <int S>
__global__ void fir_interp_generic_kernel(const typename ip_vec<S>::type* __restrict__ x, const typename ip_vec<S>::type* __restrict__ hist, int hcap,
                                          const float* __restrict__ taps, int Kp, int L, typename ip_vec<S>::type* __restrict__ y, long n_in) {
    using T = typename ip_vec<S>::type;
    const long n_out = n_in * L;
    for (long n = (long)blockIdx.x * blockDim.x + threadIdx.x; n < n_out; n += (long)gridDim.x * blockDim.x) {
        const long m = n / L;
        const int  p = (int)(n - m * L);
        T          a = T{};
        for (int q = 0; q < Kp; ++q) {
            const long pos = m - q;
            T          v   = T{};
            if (pos >= 0) v = x[pos];
            else if (pos >= -(long)hcap) v = hist[hcap + pos];
            a = ip_fma(taps[(long)q * L + p], v, a);
        }
        y[n] = a;
    }
}

// Interpolation factors that divide 16 on the f32 matrix pipe.  With G = 16 / L, 16 consecutive outputs are the L phases of G consecutive input
// positions, and the output block i (outputs 16 i .. 16 i + 15) is one column of a block-Toeplitz contraction over a sliding input window:
//     y[16 i + j] = sum_u A[j][u] B[u][i],   B[u][i] = x[G i - Hq + u],   A[j][u] = L b[(Hq + j / L - u) L + j % L]   (0 outside the taps)
// with u < 4 KS, Hq = 4 KS - G >= ceil(K / L) - 1 samples in front of the block (a multiple of G).  Only (G - 1) / (Kp + G - 1) of the products
// are padding.  The A element of lane (j, kq) at K-step ks sits at  16 + (Hq + j / L - kq) L + j % L - 4 L ks  of the zero-padded natural-order tap
// row (LDS, one ds_read per K-step shared by the wave's tiles).  The staged input is de-interleaved into G rows (row s % G, index s / G): the B
// element u = 4 ks + kq of block i is row u % G, index i + u / G -- the 16 columns of a K-step read 16 consecutive floats of one row (rows start
// 16 banks apart), at a lane base plus a wave-uniform offset.  (A linear segment has the columns G floats apart: 4-way conflicts at G = 8, and
// padding it away costs four address instructions per MFMA -- that version ran the K = 256 branches at 50 % of the matrix pipe.)  A workgroup stages
// 4096 / S input samples (+ Hq in front) and turns them into L rounds of 4096 / S outputs; each wave owns 4 / S tiles of 256 outputs per round
// (complex data: the re and the im rows under the same A, like fir_mfma_c32_kernel), D leaves as 1 KB contiguous per wave store.  The next
// segment's samples are requested before the rounds of this one.
// Bound: output writes ((4 + 4 / L) S bytes per output) at short branches, MFMA f32 (8 KS flop per real output) at long ones.
using ip_f32x4 = __attribute__((ext_vector_type(4))) float;
constexpr int kIpMaxKS = 68; // A row <= 16 + 4 KS L floats, prefetch registers sized for Hq <= 272

template <int G, int S>
__global__ __launch_bounds__(256) void fir_interp_mfma_kernel(const float* __restrict__ x, const float* __restrict__ hist, int hcap, const float* __restrict__ tab /*[16 + 4 KS L]*/, int KS,
                                                               float* __restrict__ y, long n_in, int spw /*segments per workgroup*/, float* __restrict__ new_hist, int Lrt) {
    constexpr int TPW = 4 / S, SI = 4096 / S, RIN = 64 * TPW * G, ROUNDS = 16 / G; // inputs per segment, per round
    const int     L = G == 1 ? Lrt : 16 / G; // G = 1 is also "any L": rows = 16 phases 16 blockIdx.y .. of one input position per block (rows past L idle)
    constexpr int NL = (SI + 4 * kIpMaxKS + 255) / 256;
    constexpr int LG = G == 8 ? 3 : G == 4 ? 2 : G == 2 ? 1 : 0;
    extern __shared__ float ip_sm[];
    const int Hq = 4 * KS - G, NS = SI + Hq;
    const int ROW = ((NS / G + 31) / 32) * 32 + 16; // row stride: consecutive rows start 16 banks apart
    float*    tl  = ip_sm + S * G * ROW;           // tap row
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    for (int i = tid; i < 16 + 4 * KS * L; i += 256) tl[i] = tab[i];

    // staging: a wave takes 64 consecutive samples per load; lane l takes sample (l % (64 / G)) G + l / (64 / G) of them, so that 64 / G consecutive
    // lanes write consecutive floats of one row
    const int lperm = (lane & (64 / G - 1)) * G + (lane >> (6 - LG));
    const int srow = lane >> (6 - LG), sidx = ((tid & ~63) >> LG) + (lane & (64 / G - 1)); // row, index of this lane's sample of load u = 0; + 256 u / G per load
    float     nxt[NL * S];
    auto      load_next = [&](long seg0) { // seg0 >= SI > Hq: nothing below 0; past the end of the span / of the segment the range check returns 0
        const long   i0   = seg0 - Hq;
        const long   nrec = n_in - i0 < (long)NS ? n_in - i0 : (long)NS;
        const rsrc_t r    = make_rsrc(x + i0 * S, (unsigned)(nrec > 0 ? nrec * 4 * S : 0));
        const int    vo   = ((tid & ~63) + lperm) * 4 * S;
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            if constexpr (S == 1) nxt[u] = buf_load_f(r, vo, 256 * u * 4);
            else { const float2 v = buf_load_f2(r, vo, 256 * u * 8); nxt[2 * u] = v.x; nxt[2 * u + 1] = v.y; }
        }
    };
    const long nseg = (n_in + SI - 1) / SI, sfirst = (long)blockIdx.x * spw, slast = sfirst + spw < nseg ? sfirst + spw : nseg;
    if (sfirst > 0 && sfirst < slast) load_next(sfirst * SI);
    const int  pg = G == 1 ? 16 * (int)blockIdx.y : 0;   // first phase of this workgroup's rows
    const bool arow = G > 1 || pg + col < L;              // this lane's A row exists
    const int  a0 = G == 1 ? 16 + (Hq - kq) * L + pg + col : 16 + (Hq + col / L - kq) * L + col % L; // A: j = lane & 15
    // B: u = 4 ks + kq -> row u % G, index block + u / G = [lane part] + [wave-uniform part of ks]
    const int blane = G == 1 ? kq + col : G == 2 ? (kq & 1) * ROW + (kq >> 1) + col : kq * ROW + col;
    for (long sg = sfirst; sg < slast; ++sg) {
        const long seg0 = sg * SI; // first input position of this segment
        if (sg > 0) {
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                if ((tid & ~63) + 256 * u + lperm < NS) {
                    const int ad = srow * ROW + sidx + (256 >> LG) * u;
                    if constexpr (S == 1) ip_sm[ad] = nxt[u];
                    else { ip_sm[ad] = nxt[2 * u]; ip_sm[G * ROW + ad] = nxt[2 * u + 1]; }
                }
            }
        } else {
            for (int s_ = tid; s_ < NS; s_ += 256) { // the first segment of the span reads the carried history in front of x
                const long i = s_ - Hq;
#pragma unroll
                for (int c = 0; c < S; ++c) ip_sm[(c * G + (s_ & (G - 1))) * ROW + (s_ >> LG)] = i >= 0 ? (i < n_in ? x[i * S + c] : 0.f) : (i >= -(long)hcap ? hist[(hcap + i) * S + c] : 0.f);
            }
        }
        __syncthreads();
        if (sg + 1 < slast) load_next(seg0 + SI); // in flight during the rounds below
        for (int rd = 0; rd < ROUNDS; ++rd) {
            if (seg0 + (long)rd * RIN >= n_in) break;
            ip_f32x4 acc[TPW * S];
#pragma unroll
            for (int t = 0; t < TPW * S; ++t) acc[t] = ip_f32x4{0.f, 0.f, 0.f, 0.f};
            const float* pb[TPW];
#pragma unroll
            for (int t = 0; t < TPW; ++t) pb[t] = ip_sm + blane + rd * (RIN / G) + 16 * (wave * TPW + t); // first block of tile t + this lane's column
            auto kstep = [&](int ks) {
                const float a  = arow ? tl[a0 - 4 * L * ks] : 0.f;
                const int   ko = G == 1 ? 4 * ks : G == 2 ? 2 * ks : G == 4 ? ks : 4 * (ks & 1) * ROW + (ks >> 1);
#pragma unroll
                for (int t = 0; t < TPW; ++t) {
#pragma unroll
                    for (int c = 0; c < S; ++c) acc[c * TPW + t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, pb[t][c * G * ROW + ko], acc[c * TPW + t], 0, 0, 0);
                }
            };
            int ks = 0;
            for (; ks + 4 <= KS; ks += 4) { kstep(ks); kstep(ks + 1); kstep(ks + 2); kstep(ks + 3); } // (a "#pragma unroll 4" on the runtime-count loop is refused)
            for (; ks < KS; ++ks) kstep(ks);
            // D[row = 4 kq + r][col] of tile t: output (seg0 + rd RIN) L + 16 (16 (wave TPW + t) + col) + 4 kq + r
            const long n_out = n_in * L;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                if constexpr (G == 1) { // one input position per block: outputs m L + pg + 4 kq + r, rows past L idle
                    const long m  = seg0 + (long)rd * RIN + 16 * (wave * TPW + t) + col;
                    const int  p0 = pg + 4 * kq;
                    if (m >= n_in || p0 >= L) continue;
                    const long o = m * L + p0;
                    if ((L & 3) == 0) { // 4 | L: whole, aligned groups of four rows
                        if constexpr (S == 1) *reinterpret_cast<float4*>(y + o) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
                        else {
                            float4* d = reinterpret_cast<float4*>(y + 2 * o);
                            d[0]      = make_float4(acc[t][0], acc[TPW + t][0], acc[t][1], acc[TPW + t][1]);
                            d[1]      = make_float4(acc[t][2], acc[TPW + t][2], acc[t][3], acc[TPW + t][3]);
                        }
                    } else {
                        for (int r = 0; r < 4; ++r)
                            if (p0 + r < L) {
                                if constexpr (S == 1) y[o + r] = acc[t][r];
                                else *reinterpret_cast<float2*>(y + 2 * (o + r)) = make_float2(acc[t][r], acc[TPW + t][r]);
                            }
                    }
                    continue;
                }
                const long o = (seg0 + (long)rd * RIN) * L + 16L * (16 * (wave * TPW + t) + col) + 4 * kq;
                if constexpr (S == 1) {
                    if (o + 3 < n_out) *reinterpret_cast<float4*>(y + o) = make_float4(acc[t][0], acc[t][1], acc[t][2], acc[t][3]);
                    else
                        for (int r = 0; r < 4; ++r)
                            if (o + r < n_out) y[o + r] = acc[t][r];
                } else {
                    if (o + 3 < n_out) {
                        float4* d = reinterpret_cast<float4*>(y + 2 * o);
                        d[0]      = make_float4(acc[t][0], acc[TPW + t][0], acc[t][1], acc[TPW + t][1]);
                        d[1]      = make_float4(acc[t][2], acc[TPW + t][2], acc[t][3], acc[TPW + t][3]);
                    } else {
                        for (int r = 0; r < 4; ++r)
                            if (o + r < n_out) { y[2 * (o + r)] = acc[t][r]; y[2 * (o + r) + 1] = acc[TPW + t][r]; }
                    }
                }
            }
        }
        __syncthreads(); // every wave is done with the staged segment before the next one overwrites it
    }
    if (new_hist != nullptr && blockIdx.x == 0 && blockIdx.y == 0) { // the other half of the caller's ping-pong pair: nobody reads it in this launch
        for (int h = tid; h < hcap * S; h += 256) {
            const long pos = (n_in - hcap) * S + h; // float index relative to this call's first sample
            new_hist[h]    = pos >= 0 ? x[pos] : (pos >= -(long)hcap * S ? hist[(long)hcap * S + pos] : 0.f);
        }
    }
}

template <int S>
__global__ void fir_interp_hist_kernel(const typename ip_vec<S>::type* __restrict__ x, long n_in, const typename ip_vec<S>::type* __restrict__ hold, typename ip_vec<S>::type* __restrict__ hnew, int hcap) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= hcap) return;
    const long pos = n_in - hcap + h; // stream position relative to this call's first sample
    hnew[h] = pos >= 0 ? x[pos] : (pos >= -(long)hcap ? hold[hcap + pos] : typename ip_vec<S>::type{});
}

} // namespace gr4

using namespace gr4;

struct gr4hip_fir_interp {
    int                dtype = GR4HIP_F32, S = 1;
    size_t             ntaps = 0, L = 1, Kp = 0, hcap = 32;
    std::vector<float> taps;
    DeviceBuffer       d_taps, d_hist[2], d_row; // d_row: zero-padded natural-order tap row of the matrix-pipe kernel
    int                cur = 0, KS = 0, G = 1;   // KS = 0: no matrix-pipe form for this (L, K)
    // the stream rule (common.hpp): create / reset / set_taps only note what the device state has to become; ip_state_on() enqueues it on the stream of the next call
    std::vector<float> t_host, row_host; // images of d_taps / d_row as ip_upload laid them out
    bool               taps_dirty = false, zero_hist = true;
};

constexpr size_t kIpMfmaMinOut = 32768; // shorter spans: the register-window kernel (fewer, smaller workgroups)
static bool ip_env_no_mfma() { static const bool v = std::getenv("GR4HIP_INTERP_NO_MFMA") != nullptr; return v; } // developer switch: the tests compare the two kernels

static size_t ip_bit_ceil(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }

static int ip_upload(gr4hip_fir_interp* f) {
    f->Kp = ceil_div(f->ntaps, f->L);
    std::vector<float>& t = f->t_host;
    t.assign(f->Kp * f->L, 0.f);
    for (size_t k = 0; k < f->ntaps; ++k) t[(k / f->L) * f->L + (k % f->L)] = (float)f->L * f->taps[k]; // branch p = k % L, step q = k / L, gain L
    int rc = f->d_taps.ensure(t.size() * sizeof(float)); // (growing frees the old table: hipFree waits for the device)
    if (rc) return rc;
    f->taps_dirty = true; // uploaded by ip_state_on, on the stream of the next call
    f->row_host.clear();
    f->KS = 0;
    const bool pow2 = f->L == 2 || f->L == 4 || f->L == 8 || f->L == 16;
    if (pow2 || f->L == 7 || f->L > 8) { // fir_interp_mfma_kernel: window of 4 KS >= Kp - 1 + G samples per block; (L = 3, 5, 6 fill 3 .. 6 of the 16 rows
                                         // of a G = 1 tile: measured, the register-window kernel is as fast (L = 6, 96 taps: 608 vs 596 G outputs/s) or faster (L = 5, 320 taps: 411 vs 233) there)
        const size_t G = pow2 ? 16 / f->L : 1;
        size_t       KS = ceil_div(f->Kp - 1 + G, (size_t)4);
        if (G == 8) KS += KS & 1; // the window start Hq = 4 KS - G is a whole number of G-sample blocks
        f->G = (int)G;
        if (KS <= (size_t)kIpMaxKS && 4 * KS * f->L <= 16384) { // (tap row of <= 64 KB in LDS)
            std::vector<float>& row = f->row_host;
            row.assign(16 + 4 * KS * f->L, 0.f);
            for (size_t k = 0; k < f->ntaps; ++k) row[16 + k] = (float)f->L * f->taps[k];
            rc = f->d_row.ensure(row.size() * sizeof(float));
            if (rc) return rc;
            f->KS = (int)KS;
        }
    }
    return GR4HIP_OK;
}

template <int G, int S>
static int ip_mfma_launch(const gr4hip_fir_interp* f, const void* x, void* y, long n_in, hipStream_t st, float* new_hist) {
    constexpr int SI = 4096 / S;
    const int     Hq = 4 * f->KS - G, NS = SI + Hq, ROW = ((NS / G + 31) / 32) * 32 + 16;
    const size_t  lds = ((size_t)S * G * ROW + 16 + 4 * (size_t)f->KS * f->L) * sizeof(float);
    const long    nseg = ceil_div(n_in, (long)SI);
    const int     spw = (int)std::clamp<long>(nseg / 2048, 1, 4); // >= 2048 workgroups (8 per CU) before a workgroup takes a second segment
    auto          kern = fir_interp_mfma_kernel<G, S>;
    if (lds > 48 * 1024) GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3((unsigned)ceil_div(nseg, (long)spw), G == 1 ? (unsigned)ceil_div(f->L, (size_t)16) : 1u), dim3(256), lds, st, static_cast<const float*>(x), static_cast<const float*>(f->d_hist[f->cur].ptr), (int)f->hcap,
                       static_cast<const float*>(f->d_row.ptr), f->KS, static_cast<float*>(y), n_in, spw, new_hist, (int)f->L);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}
template <int S>
static int ip_mfma_dispatch(const gr4hip_fir_interp* f, const void* x, void* y, long n_in, hipStream_t st, float* new_hist) {
    switch (f->G) {
    case 8: return ip_mfma_launch<8, S>(f, x, y, n_in, st, new_hist);
    case 4: return ip_mfma_launch<4, S>(f, x, y, n_in, st, new_hist);
    case 2: return ip_mfma_launch<2, S>(f, x, y, n_in, st, new_hist);
    default: return ip_mfma_launch<1, S>(f, x, y, n_in, st, new_hist);
    }
}
static int ip_alloc_hist(gr4hip_fir_interp* f) {
    const size_t bytes = f->hcap * f->S * sizeof(float);
    for (int k = 0; k < 2; ++k) {
        int rc = f->d_hist[k].ensure(bytes);
        if (rc) return rc;
    }
    f->zero_hist = true; // zeroed by ip_state_on, on the stream of the next call (a launch still in flight may be writing either half of the pair)
    return GR4HIP_OK;
}
// pending tap upload / zeroing of the carried history, onto the stream of the call about to be enqueued: behind this handle's earlier launches on it, in front of the next
static int ip_state_on(gr4hip_fir_interp* f, hipStream_t st) {
    if (f->taps_dirty) {
        GR4_HIP_TRY(hipMemcpyAsync(f->d_taps.ptr, f->t_host.data(), f->t_host.size() * sizeof(float), hipMemcpyHostToDevice, st));
        if (!f->row_host.empty()) GR4_HIP_TRY(hipMemcpyAsync(f->d_row.ptr, f->row_host.data(), f->row_host.size() * sizeof(float), hipMemcpyHostToDevice, st));
        f->taps_dirty = false;
    }
    if (f->zero_hist) {
        GR4_HIP_TRY(hipMemsetAsync(f->d_hist[f->cur].ptr, 0, f->hcap * f->S * sizeof(float), st));
        f->zero_hist = false;
    }
    return GR4HIP_OK;
}

template <int L, int R, int S>
static int ip_launch(const gr4hip_fir_interp* f, const void* x, void* y, long n_in, hipStream_t st) {
    using T          = typename ip_vec<S>::type;
    const size_t lds = ((f->Kp - 1) + (size_t)kIpBS * R) * sizeof(T);
    if (lds > 150 * 1024) return GR4HIP_UNSUPPORTED;
    auto kern = fir_interp_kernel<L, R, S>;
    if (lds > 48 * 1024) GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long grid = ceil_div(n_in, (long)kIpBS * R);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(kIpBS), lds, st, static_cast<const T*>(x), static_cast<const T*>(f->d_hist[f->cur].ptr), (int)f->hcap,
                       static_cast<const float*>(f->d_taps.ptr), (int)f->Kp, static_cast<T*>(y), n_in);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

template <int S>
static int ip_dispatch(const gr4hip_fir_interp* f, const void* x, void* y, long n_in, hipStream_t st) {
    using T = typename ip_vec<S>::type;
    int rc  = GR4HIP_UNSUPPORTED;
    switch (f->L) { // R L accumulators (x S): 8 .. 16 per lane
    case 2: rc = ip_launch<2, 4, S>(f, x, y, n_in, st); break;
    case 3: rc = ip_launch<3, 4, S>(f, x, y, n_in, st); break;
    case 4: rc = ip_launch<4, 2, S>(f, x, y, n_in, st); break;
    case 5: rc = ip_launch<5, 2, S>(f, x, y, n_in, st); break;
    case 6: rc = ip_launch<6, 2, S>(f, x, y, n_in, st); break;
    case 8: rc = ip_launch<8, 2, S>(f, x, y, n_in, st); break;
    default: break;
    }
    if (rc != GR4HIP_UNSUPPORTED) return rc;
    const unsigned grid = (unsigned)std::min<size_t>(ceil_div((size_t)n_in * f->L, (size_t)256), (size_t)1 << 16);
    hipLaunchKernelGGL(fir_interp_generic_kernel<S>, dim3(grid), dim3(256), 0, st, static_cast<const T*>(x), static_cast<const T*>(f->d_hist[f->cur].ptr), (int)f->hcap,
                       static_cast<const float*>(f->d_taps.ptr), (int)f->Kp, (int)f->L, static_cast<T*>(y), n_in);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

extern "C" {

int gr4hip_fir_interp_create(gr4hip_fir_interp_t** out, int dtype, const float* h_taps, size_t ntaps, size_t interp) {
    GR4_REQUIRE(out, "fir_interp: null output handle");
    GR4_REQUIRE(dtype == GR4HIP_F32 || dtype == GR4HIP_C32, "fir_interp: dtype must be F32 or C32 (got %d)", dtype);
    GR4_REQUIRE(h_taps && ntaps >= 1, "fir_interp: need at least one tap");
    GR4_REQUIRE(interp >= 1 && interp <= 4096, "fir_interp: interp must be in [1, 4096]");
    auto* f = new (std::nothrow) gr4hip_fir_interp();
    GR4_REQUIRE(f, "out of host memory");
    f->dtype = dtype;
    f->S     = dtype == GR4HIP_C32 ? 2 : 1;
    f->L     = interp;
    f->ntaps = ntaps;
    f->taps.assign(h_taps, h_taps + ntaps);
    int rc = ip_upload(f);
    if (f->Kp > f->hcap) f->hcap = ip_bit_ceil(f->Kp);
    if (!rc) rc = ip_alloc_hist(f);
    if (rc) { delete f; return rc; }
    *out = f;
    return GR4HIP_OK;
}

int gr4hip_fir_interp_set_taps(gr4hip_fir_interp_t* f, const float* h_taps, size_t ntaps) {
    GR4_REQUIRE(f && h_taps && ntaps >= 1, "fir_interp_set_taps: bad arguments");
    f->taps.assign(h_taps, h_taps + ntaps);
    f->ntaps = ntaps;
    int rc   = ip_upload(f);
    if (rc) return rc;
    if (f->Kp > f->hcap) { // like fir_filter::settingsChanged: the history is replaced (lost) only when it must grow
        f->hcap = ip_bit_ceil(f->Kp);
        return ip_alloc_hist(f);
    }
    return GR4HIP_OK;
}

int gr4hip_fir_interp_reset(gr4hip_fir_interp_t* f) {
    GR4_REQUIRE(f, "fir_interp_reset: null handle");
    return ip_alloc_hist(f);
}

int gr4hip_fir_interp_process(gr4hip_fir_interp_t* f, const void* d_in, size_t n_in, void* d_out, size_t* n_out_p, gr4hip_stream_t stream) {
    GR4_REQUIRE(f, "fir_interp_process: null handle");
    if (n_out_p) *n_out_p = n_in * f->L;
    if (n_in == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in && d_out, "fir_interp_process: null device pointer");
    hipStream_t st = as_stream(stream);
    if (const int rc = ip_state_on(f, st)) return rc;
    if (f->KS > 0 && n_in * f->L >= kIpMfmaMinOut && (uintptr_t)d_out % 16 == 0 && (uintptr_t)d_in % (4 * f->S) == 0 && !ip_env_no_mfma()) { // matrix-pipe form; writes the next history itself
        float* nh = static_cast<float*>(f->d_hist[f->cur ^ 1].ptr);
        int    rc = f->S == 1 ? ip_mfma_dispatch<1>(f, d_in, d_out, (long)n_in, st, nh) : ip_mfma_dispatch<2>(f, d_in, d_out, (long)n_in, st, nh);
        if (rc) return rc;
        f->cur ^= 1;
        return GR4HIP_OK;
    }
    int rc = f->S == 1 ? ip_dispatch<1>(f, d_in, d_out, (long)n_in, st) : ip_dispatch<2>(f, d_in, d_out, (long)n_in, st);
    if (rc) return rc;
    const unsigned hg = (unsigned)ceil_div(f->hcap, (size_t)256);
    if (f->S == 1) hipLaunchKernelGGL(fir_interp_hist_kernel<1>, dim3(hg), dim3(256), 0, st, static_cast<const float*>(d_in), (long)n_in, static_cast<const float*>(f->d_hist[f->cur].ptr), static_cast<float*>(f->d_hist[f->cur ^ 1].ptr), (int)f->hcap);
    else hipLaunchKernelGGL(fir_interp_hist_kernel<2>, dim3(hg), dim3(256), 0, st, static_cast<const float2*>(d_in), (long)n_in, static_cast<const float2*>(f->d_hist[f->cur].ptr), static_cast<float2*>(f->d_hist[f->cur ^ 1].ptr), (int)f->hcap);
    GR4_LAUNCH_CHECK();
    f->cur ^= 1;
    return GR4HIP_OK;
}

int gr4hip_fir_interp_destroy(gr4hip_fir_interp_t* f) {
    delete f;
    return GR4HIP_OK;
}

} // extern "C"
