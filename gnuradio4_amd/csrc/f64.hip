// f64.hip -- the float64 instantiations the reference registers for the hot-path blocks: fir_filter<double> and iir_filter<double, form>
// (blocks/filter/.../time_domain_filter.hpp:20, 57-60), FFT<double> (blocks/fourier/.../fft.hpp:29) and Rotator<complex<double>>
// (blocks/math/.../Rotator.hpp:15).  north_star is float32; these are the second registered type of each block, so that a graph that asks for it keeps
// running on the device.  Plain FP64 vector code (78 TFLOP/s peak), one straightforward kernel per block -- not the tuned float32 paths:
//   FIR        direct form, segment + history staged in LDS as doubles, R outputs per lane from a strided window
//   IIR        exact parallel-in-time in three passes (per-tile zero-state chunk runs + in-tile scan, one sequential walk over the tile states,
//              re-run from the true start states); every matrix power is host-precomputed in float64
//   FFT        real frames (power-of-two N <= 8192): window, half-size complex transform as radix-4 Stockham passes in LDS, 8192 / N frames per workgroup,
//              untangled into the block's outputs in double (N < 16: one frame per workgroup, radix-2 passes)
//   Rotator    closed-form phase per sample (the float64 oracle's definition up to the rounding of the accumulated sum)
#include "common.hpp"

#include <algorithm>
#include <cmath>
#include <vector>

namespace gr4 {

// ------------------------------------------------------------------------------------------------ FIR
constexpr int kF64BS = 256;

// y[m] = sum_k b[k] x[m D - k], m < n_out; xin = x with the hcap samples of history in front (virtual index -hcap .. -1 from hist).
// A lane owns R outputs 256 apart (lane-consecutive LDS reads) and walks the taps once for all of them: one broadcast tap read per R multiply-adds
template <int R>
__global__ __launch_bounds__(kF64BS) void fir64_kernel(const double* __restrict__ x, const double* __restrict__ hist, int hcap, const double* __restrict__ taps, int K, int D,
                                                       double* __restrict__ y, long n_out, long n_in) {
    extern __shared__ double smem64[];
    double*    tl  = smem64;      // [K]
    double*    xs  = smem64 + K;  // [(BS R - 1) D + K]: samples (o0 D - (K - 1)) ..
    const long o0  = (long)blockIdx.x * kF64BS * R;
    const int  len = (kF64BS * R - 1) * D + K;
    const long i0  = o0 * D - (K - 1);
    for (int i = threadIdx.x; i < K; i += kF64BS) tl[i] = taps[i];
    for (int i = threadIdx.x; i < len; i += kF64BS) {
        const long g = i0 + i;
        xs[i]        = g >= 0 ? (g < n_in ? x[g] : 0.0) : (g >= -(long)hcap ? hist[hcap + g] : 0.0);
    }
    __syncthreads();
    const double* w[R];
    double        acc[R];
#pragma unroll
    for (int r = 0; r < R; ++r) {
        w[r]   = xs + (long)(r * kF64BS + threadIdx.x) * D + (K - 1);
        acc[r] = 0.0;
    }
    for (int k = 0; k < K; ++k) {
        const double t = tl[k];
#pragma unroll
        for (int r = 0; r < R; ++r) acc[r] = fma(t, w[r][-k], acc[r]);
    }
#pragma unroll
    for (int r = 0; r < R; ++r) {
        const long o = o0 + (long)r * kF64BS + threadIdx.x;
        if (o < n_out) y[o] = acc[r];
    }
}

// fir_filter<double>, no decimation, 17 .. 2048 taps, long spans: the block-Toeplitz contraction of fir_batched.hip on v_mfma_f64_16x16x4_f64 (the FP64 VALU's
// rate, reached from one wave per SIMD with one operand read per 2048 flop instead of one LDS read per fma).  With Kp = taps rounded up to 64 / 128 / 256 (compile-time K-loop) or to the next multiple of 64 (run-time K-loop):
//     y[16 i + j] = sum_{u < Kp + 16} A[j][u] B[u][i],   A[j][u] = b[Kp + j - u]  (Toeplitz: read from the tap row in LDS),   B[u][i] = x[16 i - Kp + u]
// 4096 outputs per segment (4 waves x 4 tiles of 256), the segment + Kp samples staged 18 / 16-padded (8-byte elements: a K-step's 32 lanes hit 64 different
// banks), the next segment requested into registers before the MFMAs of this one, workgroup 0 writes the next history.  Bound: MFMA f64, 2 (Kp + 16) flop per sample.
using f64x4 = __attribute__((ext_vector_type(4))) double;
constexpr int kF64Seg = 4096, kF64SegPerWg = 4;
template <int KSC> // K-steps of 4 (Kp = 4 KS - 16); 0: given at run time (Kp = 320 .. 2048 in steps of 64)
__global__ __launch_bounds__(256) void fir64_mfma_kernel(const double* __restrict__ x, const double* __restrict__ hist, int hcap, const double* __restrict__ trow /*[Kp + 32]: index 16 + q = b[q]*/,
                                                          double* __restrict__ y, long n, double* __restrict__ new_hist, int ks_rt) {
    const int     KS = KSC ? KSC : ks_rt, Kp = 4 * KS - 16, NPAD = (kF64Seg + Kp) / 16 * 18;
    constexpr int NL = KSC ? (kF64Seg + 4 * KSC - 16 + 255) / 256 : (kF64Seg + 2048 + 255) / 256;
    extern __shared__ double smem64[];
    double*   xs = smem64;        // [NPAD]
    double*   tp = smem64 + NPAD; // [Kp + 32]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
    for (int i = tid; i < Kp + 32; i += 256) tp[i] = trow[i];
    // v_mfma_f64_16x16x4_f64 returns D[4 r + kq][col] in register r of lane (col, kq) (the f32 form: D[4 kq + r][col]).  A's row i' therefore carries the
    // taps of output j = 4 (i' & 3) + (i' >> 2): register r of a lane is then output 4 kq + r of its block, and a lane's four results are contiguous
    const double* pa = tp + 16 + Kp + (4 * (col & 3) + (col >> 2)) - kq;
    double        nxt[NL];
    auto          load_next = [&](long seg0) { // seg0 >= kF64Seg > Kp: nothing below 0; past the end of the span / of the segment: 0
        const long i0 = seg0 - Kp;
#pragma unroll
        for (int u = 0; u < NL; ++u) {
            const long i = i0 + tid + 256 * u;
            nxt[u]       = (tid + 256 * u < kF64Seg + Kp && i < n) ? x[i] : 0.0;
        }
    };
    const long nseg = (n + kF64Seg - 1) / kF64Seg, sfirst = (long)blockIdx.x * kF64SegPerWg, slast = sfirst + kF64SegPerWg < nseg ? sfirst + kF64SegPerWg : nseg;
    if (sfirst > 0 && sfirst < slast) load_next(sfirst * kF64Seg);
    for (long sg = sfirst; sg < slast; ++sg) {
        const long seg0 = sg * kF64Seg;
        if (sg > 0) {
#pragma unroll
            for (int u = 0; u < NL; ++u) {
                const int s_ = tid + 256 * u;
                if (s_ < kF64Seg + Kp) xs[s_ + 2 * (s_ >> 4)] = nxt[u];
            }
        } else {
            for (int s_ = tid; s_ < kF64Seg + Kp; s_ += 256) { // the first segment of the span reads the carried history in front of x
                const long i           = s_ - Kp;
                xs[s_ + 2 * (s_ >> 4)] = i >= 0 ? (i < n ? x[i] : 0.0) : (i >= -(long)hcap ? hist[hcap + i] : 0.0);
            }
        }
        __syncthreads();
        if (sg + 1 < slast) load_next(seg0 + kF64Seg); // in flight during the MFMAs below
#pragma unroll
        for (int pair = 0; pair < 2; ++pair) { // two independent accumulators per round
            const int     ib0 = 16 * (4 * wave + 2 * pair), ib1 = ib0 + 16;
            f64x4         acc0 = {0.0, 0.0, 0.0, 0.0}, acc1 = acc0;
            const double* p0 = xs + 18 * (ib0 + col) + kq;
            const double* p1 = xs + 18 * (ib1 + col) + kq;
            auto kstep = [&](int ks) {
                const int    off = 4 * ks + 2 * (ks >> 2);
                const double a   = pa[-4 * ks];
                acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, p0[off], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, p1[off], acc1, 0, 0, 0);
            };
            if constexpr (KSC != 0) {
#pragma unroll
                for (int ks = 0; ks < KSC; ++ks) kstep(ks);
            } else {
                for (int ks = 0; ks < KS; ks += 4) { kstep(ks); kstep(ks + 1); kstep(ks + 2); kstep(ks + 3); } // KS = Kp / 4 + 4 with 64 | Kp: a multiple of 4
            }
            // D[row = 4 kq + r][col]: y[16 (ib + col) + 4 kq + r]
            const long o0 = seg0 + 16L * (ib0 + col) + 4 * kq, o1 = seg0 + 16L * (ib1 + col) + 4 * kq;
            if (o0 + 3 < n) {
                *reinterpret_cast<double2*>(y + o0)     = make_double2(acc0[0], acc0[1]);
                *reinterpret_cast<double2*>(y + o0 + 2) = make_double2(acc0[2], acc0[3]);
            } else
                for (int r = 0; r < 4; ++r)
                    if (o0 + r < n) y[o0 + r] = acc0[r];
            if (o1 + 3 < n) {
                *reinterpret_cast<double2*>(y + o1)     = make_double2(acc1[0], acc1[1]);
                *reinterpret_cast<double2*>(y + o1 + 2) = make_double2(acc1[2], acc1[3]);
            } else
                for (int r = 0; r < 4; ++r)
                    if (o1 + r < n) y[o1 + r] = acc1[r];
        }
        __syncthreads(); // every wave is done with the staged segment before the next one overwrites it
    }
    if (blockIdx.x == 0) { // the other half of the caller's ping-pong pair: nobody reads it in this launch
        for (int h = tid; h < hcap; h += 256) {
            const long g = n - hcap + h;
            new_hist[h]  = g >= 0 ? x[g] : hist[hcap + g];
        }
    }
}

// new_hist[h] = virtual sample n_in - hcap + h of (old_hist ++ x)
__global__ void hist64_kernel(const double* __restrict__ x, const double* __restrict__ old_hist, double* __restrict__ new_hist, long n_in, int hcap) {
    const int h = blockIdx.x * blockDim.x + threadIdx.x;
    if (h >= hcap) return;
    const long g = n_in - hcap + h;
    new_hist[h]  = g >= 0 ? x[g] : old_hist[hcap + g];
}

// ------------------------------------------------------------------------------------------------ IIR
constexpr int kI64L  = 32;  // samples per lane chunk
constexpr int kI64BS = 256; // chunks per tile
constexpr int kI64MP = 8;   // state dimension the kernels are built for (up to 4 biquads, or one section of order <= 8)
constexpr int kI64P  = kI64L + 1; // pitch of a lane's chunk in the staged tile: 33 doubles = 66 dwords, a lane group's 8-byte accesses hit 64 different banks
static_assert(kI64L == 32, "the staging index arithmetic below uses idx >> 5, idx & 31");

struct Iir64Desc {
    int    nsec, ord;              // sections of order `ord` (state: ord values per section, direct form II)
    double b[kI64MP / 1][9];       // [section][ord + 1]   (at most 8 sections of order 1)
    double a[kI64MP / 1][9];
};

// one sample through the cascade (computeFilter, time_domain_filter.hpp / FilterTool.hpp DF-II): s = [section][ord] delay lines.  The section count and
// order are template parameters: every loop unrolls and the state stays in registers (as run-time loops the state arrays lived in scratch memory: 8x slower)
template <int NSEC, int ORD>
__device__ __forceinline__ double iir64_step(const Iir64Desc& d, double (&s)[NSEC * ORD], double x) {
#pragma unroll
    for (int c = 0; c < NSEC; ++c) {
        double w = x;
#pragma unroll
        for (int k = 1; k <= ORD; ++k) w = fma(-d.a[c][k], s[c * ORD + k - 1], w);
        double yv = d.b[c][0] * w;
#pragma unroll
        for (int k = 1; k <= ORD; ++k) yv = fma(d.b[c][k], s[c * ORD + k - 1], yv);
#pragma unroll
        for (int k = ORD - 1; k > 0; --k) s[c * ORD + k] = s[c * ORD + k - 1];
        s[c * ORD] = w;
        x          = yv;
    }
    return x;
}
template <int MP>
__device__ __forceinline__ void matvec64(const double* __restrict__ M, const double (&v)[MP], double (&out)[MP]) { // out = M v (row-major MP x MP)
#pragma unroll
    for (int i = 0; i < MP; ++i) {
        double acc = 0.0;
#pragma unroll
        for (int j = 0; j < MP; ++j) acc = fma(M[i * MP + j], v[j], acc);
        out[i] = acc;
    }
}

// The tile's 8192 samples go through LDS: consecutive lanes load consecutive samples (a lane reading its own 256-byte chunk straight from global memory
// touches 64 cache lines per load instruction -- the first version ran 4 biquads at 26 Gsamples/s), then every lane walks its chunk at pitch 33.
__device__ __forceinline__ void iir64_stage_tile(double* sm, const double* __restrict__ x, long tile, long n, int l) {
    const long t0 = tile * (long)(kI64BS * kI64L);
#pragma unroll 8
    for (int j = 0; j < kI64L; ++j) {
        const int idx = j * kI64BS + l;
        sm[(idx >> 5) * kI64P + (idx & 31)] = t0 + idx < n ? x[t0 + idx] : 0.0;
    }
}

// pass Z: per tile of 256 chunks: zero-state end state of every chunk, inclusive scan over the chunks (P_l = state at the end of chunk l when the tile
// starts from zero), all P_l to scratch, the tile's own end state Z_t = P_255
template <int NSEC, int ORD>
__global__ __launch_bounds__(kI64BS) void iir64_pass_z(Iir64Desc d, const double* __restrict__ x, long n, const double* __restrict__ phiPow /*[8][MP][MP]: Phi_L^(2^k)*/,
                                                       double* __restrict__ P /*[tiles][256][MP]*/, double* __restrict__ Z /*[tiles][MP]*/) {
    constexpr int MP = NSEC * ORD;
    extern __shared__ double smem64[];
    double (*sc)[MP][kI64BS] = reinterpret_cast<double (*)[MP][kI64BS]>(smem64 + kI64BS * kI64P); // [2][MP][256], lane-contiguous
    const long tile = blockIdx.x;
    const int  l    = threadIdx.x;
    iir64_stage_tile(smem64, x, tile, n, l);
    __syncthreads();
    double        s[MP];
    const double* xc = smem64 + l * kI64P;
#pragma unroll
    for (int i = 0; i < MP; ++i) s[i] = 0.0;
#pragma unroll 8
    for (int i = 0; i < kI64L; ++i) (void)iir64_step<NSEC, ORD>(d, s, xc[i]); // (the zero padding of the last tile only moves states nobody reads)
    int cur = 0;
#pragma unroll
    for (int i = 0; i < MP; ++i) sc[0][i][l] = s[i];
    __syncthreads();
    for (int k = 0, off = 1; off < kI64BS; ++k, off <<= 1) { // Hillis-Steele: P_l <- Phi_L^off P_{l - off} + P_l
        double v[MP];
#pragma unroll
        for (int i = 0; i < MP; ++i) v[i] = sc[cur][i][l];
        if (l >= off) {
            double u[MP], t[MP];
#pragma unroll
            for (int i = 0; i < MP; ++i) u[i] = sc[cur][i][l - off];
            matvec64<MP>(phiPow + (long)k * MP * MP, u, t);
#pragma unroll
            for (int i = 0; i < MP; ++i) v[i] += t[i];
        }
#pragma unroll
        for (int i = 0; i < MP; ++i) sc[cur ^ 1][i][l] = v[i];
        cur ^= 1;
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < MP; ++i) P[(tile * kI64BS + l) * MP + i] = sc[cur][i][l];
    if (l == kI64BS - 1) {
#pragma unroll
        for (int i = 0; i < MP; ++i) Z[tile * MP + i] = sc[cur][i][l];
    }
}

// pass B: the tiles in order, T_{t+1} = Phi_B T_t + Z_t; T_0 = the carried state -- the same recurrence one level up, evaluated the same way: lane l owns
// kI64G consecutive tiles, runs them from zero (lane 0: from the carry), a Hillis-Steele scan over the 256 lanes with Phi_B^(G 2^k) gives every lane the state
// its group starts from, and a second run writes the T_t.  4096 tiles (2^25 samples) per round; longer spans take further rounds from the carried state.
// (One lane walking all tiles in order was 68 % of the cascade's time at 2^24 samples: 0.52 of 0.76 ms.)
constexpr int kI64G = 16; // tiles per lane
template <int MP>
__global__ __launch_bounds__(256) void iir64_pass_b(const double* __restrict__ phiB /*[9][MP][MP]: Phi_B, then Phi_B^(G 2^k), k < 8*/, const double* __restrict__ Z, long tiles,
                                                     const double* __restrict__ state0, double* __restrict__ T /*[tiles][MP]*/) {
    __shared__ double sc[2][MP][256], carry[MP];
    const int l = threadIdx.x;
    if (l < MP) carry[l] = state0[l];
    __syncthreads();
    for (long t0 = 0; t0 < tiles; t0 += 256 * kI64G) {
        const long g0 = t0 + (long)l * kI64G;
        double     s[MP], t[MP];
#pragma unroll
        for (int i = 0; i < MP; ++i) s[i] = l == 0 ? carry[i] : 0.0;
        for (int c = 0; c < kI64G; ++c) { // end state of the group (tiles past the end contribute nothing: nobody reads what follows them)
            matvec64<MP>(phiB, s, t);
#pragma unroll
            for (int i = 0; i < MP; ++i) s[i] = t[i] + (g0 + c < tiles ? Z[(g0 + c) * MP + i] : 0.0);
        }
        int cur = 0;
#pragma unroll
        for (int i = 0; i < MP; ++i) sc[0][i][l] = s[i];
        __syncthreads();
        for (int k = 0, off = 1; off < 256; ++k, off <<= 1) { // P_l <- Phi_B^(G off) P_{l - off} + P_l
            double v[MP];
#pragma unroll
            for (int i = 0; i < MP; ++i) v[i] = sc[cur][i][l];
            if (l >= off) {
                double u[MP];
#pragma unroll
                for (int i = 0; i < MP; ++i) u[i] = sc[cur][i][l - off];
                matvec64<MP>(phiB + (long)(1 + k) * MP * MP, u, t);
#pragma unroll
                for (int i = 0; i < MP; ++i) v[i] += t[i];
            }
#pragma unroll
            for (int i = 0; i < MP; ++i) sc[cur ^ 1][i][l] = v[i];
            cur ^= 1;
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < MP; ++i) s[i] = l == 0 ? carry[i] : sc[cur][i][l - 1]; // the state this lane's group starts from
        __syncthreads(); // (lane 0 has read the carry)
        if (l == 255) {
#pragma unroll
            for (int i = 0; i < MP; ++i) carry[i] = sc[cur][i][255]; // only read when another round follows, i.e. when every group was full
        }
        for (int c = 0; c < kI64G && g0 + c < tiles; ++c) {
#pragma unroll
            for (int i = 0; i < MP; ++i) T[(g0 + c) * MP + i] = s[i];
            matvec64<MP>(phiB, s, t);
#pragma unroll
            for (int i = 0; i < MP; ++i) s[i] = t[i] + Z[(g0 + c) * MP + i];
        }
        __syncthreads();
    }
}

// pass Y: chunk l of tile t starts from P_{l-1} + Phi_L^l T_t; re-run, write y; the lane that holds the last sample of the span writes the state after it
template <int NSEC, int ORD>
__global__ __launch_bounds__(kI64BS) void iir64_pass_y(Iir64Desc d, const double* __restrict__ x, long n, const double* __restrict__ phiL /*[256][MP][MP]: Phi_L^l*/,
                                                       const double* __restrict__ P, const double* __restrict__ T, double* __restrict__ y, double* __restrict__ state_out) {
    constexpr int MP = NSEC * ORD;
    extern __shared__ double smem64[];
    const long tile = blockIdx.x;
    const int  l    = threadIdx.x;
    const long i0   = (tile * kI64BS + l) * kI64L;
    iir64_stage_tile(smem64, x, tile, n, l);
    __syncthreads();
    double* xc = smem64 + l * kI64P; // y takes the place of x, sample by sample
    double  s[MP], tt[MP], t0[MP];
#pragma unroll
    for (int i = 0; i < MP; ++i) t0[i] = T[tile * MP + i];
    matvec64<MP>(phiL + (long)l * MP * MP, t0, tt);
#pragma unroll
    for (int i = 0; i < MP; ++i) s[i] = tt[i] + (l > 0 ? P[(tile * kI64BS + l - 1) * MP + i] : 0.0);
    if (i0 + kI64L <= n) {
#pragma unroll 8
        for (int i = 0; i < kI64L; ++i) xc[i] = iir64_step<NSEC, ORD>(d, s, xc[i]);
    } else {
        for (int i = 0; i < kI64L && i0 + i < n; ++i) xc[i] = iir64_step<NSEC, ORD>(d, s, xc[i]);
    }
    if (i0 < n && i0 + kI64L >= n) { // the lane that holds the last sample of the span: the state after it
#pragma unroll
        for (int i = 0; i < MP; ++i) state_out[i] = s[i];
    }
    __syncthreads();
    const long t0s = tile * (long)(kI64BS * kI64L);
#pragma unroll 4
    for (int j = 0; j < kI64L; ++j) { // coalesced: consecutive lanes store consecutive samples
        const int idx = j * kI64BS + l;
        if (t0s + idx < n) y[t0s + idx] = smem64[(idx >> 5) * kI64P + (idx & 31)];
    }
}

// the three passes for one (section count, order) shape
template <int NSEC, int ORD>
static int iir64_run(const Iir64Desc& d, const double* x, long n, long tiles, const double* phiPow, const double* phiL, const double* phiB, const double* state_in, double* state_out, double* P,
                     double* Z, double* T, double* y, hipStream_t st) {
    constexpr size_t ldsY = (size_t)kI64BS * kI64P * sizeof(double), ldsZ = ldsY + (size_t)2 * NSEC * ORD * kI64BS * sizeof(double);
    GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(iir64_pass_z<NSEC, ORD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsZ));
    GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(iir64_pass_y<NSEC, ORD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsY));
    hipLaunchKernelGGL((iir64_pass_z<NSEC, ORD>), dim3((unsigned)tiles), dim3(kI64BS), ldsZ, st, d, x, n, phiPow, P, Z);
    GR4_LAUNCH_CHECK();
    hipLaunchKernelGGL((iir64_pass_b<NSEC * ORD>), dim3(1), dim3(256), 0, st, phiB, (const double*)Z, tiles, state_in, T);
    GR4_LAUNCH_CHECK();
    hipLaunchKernelGGL((iir64_pass_y<NSEC, ORD>), dim3((unsigned)tiles), dim3(kI64BS), ldsY, st, d, x, n, phiL, (const double*)P, (const double*)T, y, state_out);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

// ------------------------------------------------------------------------------------------------ FFT (real double frames)
struct Fft64Out {
    double *mag, *phase, *re, *im;
    int     in_db, in_deg, unwrap;
};
// one workgroup per frame: x w -> complex points in LDS (bit-reversed), in-place radix-2 passes, outputs as FFT<T>::processBulk
// for real input (fft.hpp:147-171, 221-227: magnitude / phase of bins 0 .. N/2-1, Re / Im of bins N/2 .. N-1)
__global__ __launch_bounds__(256) void fft64_kernel(const double* __restrict__ x, const double* __restrict__ win, const double2* __restrict__ tw /*[N/2]: W_N^j*/, int log2n, long n_frames,
                                                    Fft64Out o) {
    extern __shared__ double2 sm2[];
    const int  N = 1 << log2n, half = N >> 1;
    double2*   A = sm2; // the whole frame as complex points, in place: 8192 x 16 B = 128 KiB
    const long f = blockIdx.x;
    for (int i = threadIdx.x; i < N; i += 256) { // bit-reversed load, then in-place decimation-in-time butterflies
        const double v                      = x[f * N + i] * (win ? win[i] : 1.0);
        A[__brev((unsigned)i) >> (32 - log2n)] = make_double2(v, 0.0);
    }
    __syncthreads();
    for (int s = 1, hm = 1; s <= log2n; ++s, hm <<= 1) { // hm = half the butterfly span of this pass
        for (int j = threadIdx.x; j < half; j += 256) {
            const int     k    = j & (hm - 1);
            const int     base = ((j - k) << 1) + k;
            const double2 w    = tw[k * (half / hm)]; // W_{2 hm}^k
            const double2 u    = A[base], b = A[base + hm];
            const double2 t    = make_double2(fma(b.x, w.x, -b.y * w.y), fma(b.x, w.y, b.y * w.x));
            A[base]            = make_double2(u.x + t.x, u.y + t.y);
            A[base + hm]       = make_double2(u.x - t.x, u.y - t.y);
        }
        __syncthreads();
    }
    const double pi = 3.14159265358979323846;
    for (int k = threadIdx.x; k < half; k += 256) {
        const double2 X = A[k], Xh = A[k + half];
        if (o.re) o.re[f * half + k] = Xh.x;
        if (o.im) o.im[f * half + k] = Xh.y;
        if (o.mag) {
            double m = hypot(X.x, X.y) * 2.0 / (double)N;
            if (o.in_db) m = m > 0.0 ? 20.0 * log10(m) : -1.7976931348623157e308;
            o.mag[f * half + k] = m;
        }
        if (o.phase) {
            double ph = atan2(X.y, X.x);
            if (o.in_deg && !o.unwrap) ph = ph * 180.0 / pi;
            o.phase[f * half + k] = ph;
        }
    }
    if (o.phase && o.unwrap) { // fft_common.hpp:71-89 unwrapPhase: sequential over the half spectrum (one lane; the frame's phases are in global memory)
        __syncthreads();
        if (threadIdx.x == 0) {
            double* ph   = o.phase + f * half;
            double  corr = 0.0, prev = ph[0];
            for (int k = 1; k < half; ++k) {
                const double raw = ph[k], diff = raw - prev;
                if (diff > pi) corr -= 2.0 * pi;
                else if (diff < -pi) corr += 2.0 * pi;
                prev  = raw;
                ph[k] = raw + corr;
            }
            if (o.in_deg)
                for (int k = 0; k < half; ++k) ph[k] = ph[k] * 180.0 / pi;
        }
    }
}

// The same block for N >= 16, the way the float kernels do it: a real frame of N points is HALF a complex transform -- z[n] = x[2n] + i x[2n+1], Z = FFT_M(z), M = N/2,
// X[k] = (Z[k] + conj Z[M-k]) / 2 - i W_N^k (Z[k] - conj Z[M-k]) / 2 -- and the transform runs as Stockham autosort passes of radix 4 (one radix-2 pass in front when
// log2 M is odd) on 4096 complex points per workgroup = 8192 / N frames side by side: every lane does four butterflies per pass whatever N is, reads at the constant
// stride M / 4, scatters into natural order (no bit reversal), two LDS barriers per pass.  One 64 KiB buffer, two workgroups per CU.
constexpr int kF64Pts = 4096; // complex points per workgroup
__device__ __forceinline__ double2 c64_mul(double2 a, double2 b) { return make_double2(fma(a.x, b.x, -a.y * b.y), fma(a.x, b.y, a.y * b.x)); }
__device__ __forceinline__ double2 c64_tw(const double2* __restrict__ tw, int j, int M) { // W_{2M}^j for j < 2M from the half-period table
    const double2 w = tw[j & (M - 1)];
    return (j & M) ? make_double2(-w.x, -w.y) : w;
}
__global__ __launch_bounds__(256, 2) void fft64_r2c_kernel(const double* __restrict__ x, const double* __restrict__ win, const double2* __restrict__ tw /*[N/2]: W_N^j*/, int log2n, long n_frames,
                                                           Fft64Out o) {
    extern __shared__ double2 sm2[];
    const int  N = 1 << log2n, M = N >> 1, log2m = log2n - 1;
    const int  fpb = kF64Pts / M;                 // frames per workgroup
    const long f0  = (long)blockIdx.x * fpb;
    const int  tid = threadIdx.x;
    double2*   A   = sm2;
    // load: complex point p of the workgroup = real samples 2p, 2p + 1 of the frames' concatenation (16-byte loads, lane-contiguous)
    for (int p = tid; p < kF64Pts; p += 256) {
        const long g = f0 * N + 2L * p;
        double2    v = make_double2(0.0, 0.0);
        if (g < n_frames * N) {
            v = *reinterpret_cast<const double2*>(x + g);
            if (win) { const int i = (2 * p) & (N - 1); v.x *= win[i]; v.y *= win[i + 1]; }
        }
        A[p] = v;
    }
    __syncthreads();
    const int Mq = M >> 2;
    int       Ns = 1;
    if (log2m & 1) { // radix 2 first: butterflies (j, j + M/2) of every frame, no twiddles
        double2 u[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { const int bj = tid + 256 * i, fr = bj / (M >> 1), j = bj % (M >> 1); u[i] = A[fr * M + j]; b[i] = A[fr * M + j + (M >> 1)]; }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const int bj = tid + 256 * i, fr = bj / (M >> 1), j = bj % (M >> 1);
            A[fr * M + 2 * j]     = make_double2(u[i].x + b[i].x, u[i].y + b[i].y);
            A[fr * M + 2 * j + 1] = make_double2(u[i].x - b[i].x, u[i].y - b[i].y);
        }
        __syncthreads();
        Ns = 2;
    }
    for (; Ns < M; Ns <<= 2) {
        double2 v[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int bj = tid + 256 * i, fr = bj / Mq, j = bj % Mq, k = j & (Ns - 1);
            const int step = M / (2 * Ns) * k; // W_{4 Ns}^k = W_{2M}^{k 2M / (4 Ns)}
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const double2 a = A[fr * M + j + r * Mq];
                v[i][r]         = r == 0 ? a : c64_mul(a, c64_tw(tw, step * r, M));
            }
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int     bj = tid + 256 * i, fr = bj / Mq, j = bj % Mq, k = j & (Ns - 1);
            const int     j0 = ((j - k) << 2) + k; // (j / Ns) 4 Ns + k
            const double2 t0 = make_double2(v[i][0].x + v[i][2].x, v[i][0].y + v[i][2].y), t1 = make_double2(v[i][0].x - v[i][2].x, v[i][0].y - v[i][2].y);
            const double2 t2 = make_double2(v[i][1].x + v[i][3].x, v[i][1].y + v[i][3].y), t3 = make_double2(v[i][1].y - v[i][3].y, v[i][3].x - v[i][1].x); // -i (v1 - v3)
            double2*      d  = A + fr * M + j0;
            d[0]             = make_double2(t0.x + t2.x, t0.y + t2.y);
            d[Ns]            = make_double2(t1.x + t3.x, t1.y + t3.y);
            d[2 * Ns]        = make_double2(t0.x - t2.x, t0.y - t2.y);
            d[3 * Ns]        = make_double2(t1.x - t3.x, t1.y - t3.y);
        }
        __syncthreads();
    }
    // untangle + the block's outputs (fft.hpp:147-171, 221-227): magnitude / phase of bins 0 .. N/2-1, Re / Im of bins N/2 .. N-1 (= conj X[N - k])
    const double pi = 3.14159265358979323846;
    for (int p = tid; p < kF64Pts; p += 256) {
        const int  fr = p / M, k = p % M;
        const long f  = f0 + fr;
        if (f >= n_frames) continue;
        const double2 Z = A[fr * M + k], Zc = A[fr * M + ((M - k) & (M - 1))];
        const double2 e = make_double2(0.5 * (Z.x + Zc.x), 0.5 * (Z.y - Zc.y)), dd = make_double2(0.5 * (Z.x - Zc.x), 0.5 * (Z.y + Zc.y)); // (Z + conj Zc) / 2, (Z - conj Zc) / 2
        const double2 w = tw[k];                                                                                                           // W_N^k
        const double2 t = c64_mul(dd, w);
        const double2 X = make_double2(e.x + t.y, e.y - t.x); // e - i t
        if (k == 0) { // bin N/2 = Re Z0 - Im Z0 (real)
            if (o.re) o.re[f * M] = Z.x - Z.y;
            if (o.im) o.im[f * M] = 0.0;
        } else { // bin N - k = conj X[k]
            if (o.re) o.re[f * M + (M - k)] = X.x;
            if (o.im) o.im[f * M + (M - k)] = -X.y;
        }
        if (o.mag) {
            double m = hypot(X.x, X.y) * 2.0 / (double)N;
            if (o.in_db) m = m > 0.0 ? 20.0 * log10(m) : -1.7976931348623157e308;
            o.mag[f * M + k] = m;
        }
        if (o.phase) {
            double ph = atan2(X.y, X.x);
            if (o.in_deg && !o.unwrap) ph = ph * 180.0 / pi;
            o.phase[f * M + k] = ph;
        }
    }
    if (o.phase && o.unwrap) { // fft_common.hpp:71-89 unwrapPhase: sequential over the half spectrum (one lane per frame; the phases are in global memory)
        __syncthreads();
        for (int fr = tid; fr < fpb && f0 + fr < n_frames; fr += 256) {
            double* ph   = o.phase + (f0 + fr) * M;
            double  corr = 0.0, prev = ph[0];
            for (int k = 1; k < M; ++k) {
                const double raw = ph[k], diff = raw - prev;
                if (diff > pi) corr -= 2.0 * pi;
                else if (diff < -pi) corr += 2.0 * pi;
                prev  = raw;
                ph[k] = raw + corr;
            }
            if (o.in_deg)
                for (int k = 0; k < M; ++k) ph[k] = ph[k] * 180.0 / pi;
        }
    }
}

// ------------------------------------------------------------------------------------------------ Rotator<complex<double>>
// y[i] = x[i] e^{i (phase0 + (i + 1) inc)} (Rotator.hpp:51-61 with the accumulated phase in closed form); the index is split so that both products are exact
__global__ __launch_bounds__(256) void rotator64_kernel(const double2* __restrict__ x, double2* __restrict__ y, long n, double phase0, double inc) {
    const double two_pi = 6.283185307179586476925286766559;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const long   k  = i + 1;
        const double hi = (double)(k >> 20) * 1048576.0, lo = (double)(k & 1048575);
        double       ph = fma(hi, inc, phase0);
        ph              = ph - two_pi * floor(ph / two_pi);
        ph              = fma(lo, inc, ph);
        double s, c;
        sincos(ph, &s, &c);
        const double2 v = x[i];
        y[i]            = make_double2(fma(v.x, c, -v.y * s), fma(v.x, s, v.y * c));
    }
}

} // namespace gr4

using namespace gr4;

// ================================================================================================ handles
struct gr4hip_fir64 {
    std::vector<double> taps;
    size_t              decim = 1, hcap = 32;
    DeviceBuffer        d_taps, d_hist[2], d_trow; // d_trow: zero-padded tap row of the matrix-pipe kernel
    int                 cur = 0;
    // the stream rule (common.hpp): create / reset / set_taps only note what the device state has to become; fir64_state_on() enqueues it on the stream of the next call
    std::vector<double> row_host;
    bool                taps_dirty = false, zero_hist = true;
};
struct gr4hip_iir64 {
    Iir64Desc    d{};
    int          mp = 0;
    DeviceBuffer d_phiPow, d_phiL, d_phiB, d_state[2], d_P, d_Z, d_T;
    int          cur = 0;
    bool         zero_state = true; // create / reset note it, the next call zeroes the state pair on its own stream
};
struct gr4hip_fft64 {
    size_t       N = 0;
    int          log2n = 0, flags = 0;
    DeviceBuffer d_win, d_tw;
    bool         windowed = false;
};
struct gr4hip_rotator64 {
    double inc = 0.0, phase = 0.0, initial = 0.0;
};

static size_t bit_ceil_sz(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }

static int fir64_upload(gr4hip_fir64* f, const double* taps, size_t ntaps, bool keep_history) {
    const size_t hcap = std::max<size_t>(32, bit_ceil_sz(ntaps)); // HistoryBuffer{32}, grown by settingsChanged (time_domain_filter.hpp:36-42)
    int          rc   = f->d_taps.ensure(ntaps * sizeof(double)); // (growing frees the old table: hipFree waits for the device)
    if (rc) return rc;
    { // fir64_mfma_kernel: row[16 + q] = b[q], zero elsewhere
        const size_t         Kp = ntaps <= 64 ? 64 : ntaps <= 128 ? 128 : ntaps <= 256 ? 256 : (ntaps + 63) / 64 * 64;
        std::vector<double>& row = f->row_host;
        row.assign(Kp + 32, 0.0);
        for (size_t q = 0; q < ntaps; ++q) row[16 + q] = taps[q];
        rc = f->d_trow.ensure(row.size() * sizeof(double));
        if (rc) return rc;
    }
    if (!keep_history || hcap != f->hcap || !f->d_hist[0].ptr) { // a history that has to grow starts empty, as upstream's replaced HistoryBuffer does
        for (auto& h : f->d_hist) {
            rc = h.ensure(hcap * sizeof(double));
            if (rc) return rc;
        }
        f->zero_hist = true;
    }
    f->hcap = hcap;
    f->taps.assign(taps, taps + ntaps);
    f->taps_dirty = true; // uploaded by fir64_state_on, on the stream of the next call
    return GR4HIP_OK;
}
// pending tap upload / zeroing of the carried history, onto the stream of the call about to be enqueued: behind this handle's earlier launches on it, in front of the next
static int fir64_state_on(gr4hip_fir64* f, hipStream_t st) {
    if (f->taps_dirty) {
        GR4_HIP_TRY(hipMemcpyAsync(f->d_taps.ptr, f->taps.data(), f->taps.size() * sizeof(double), hipMemcpyHostToDevice, st));
        GR4_HIP_TRY(hipMemcpyAsync(f->d_trow.ptr, f->row_host.data(), f->row_host.size() * sizeof(double), hipMemcpyHostToDevice, st));
        f->taps_dirty = false;
    }
    if (f->zero_hist) {
        GR4_HIP_TRY(hipMemsetAsync(f->d_hist[f->cur].ptr, 0, f->hcap * sizeof(double), st));
        f->zero_hist = false;
    }
    return GR4HIP_OK;
}

extern "C" {

// ---------------------------------------------------------------- fir_filter<double>
int gr4hip_fir64_create(gr4hip_fir64_t** out, const double* h_taps, size_t ntaps, size_t decim) {
    GR4_REQUIRE(out && h_taps && ntaps >= 1, "fir64: taps required");
    GR4_REQUIRE(decim >= 1, "fir64: decim must be >= 1");
    if (ntaps > 2048 || decim > 32) { set_error("fir64: ntaps=%zu decim=%zu is outside the float64 kernel (<= 2048 taps, decim <= 32)", ntaps, decim); return GR4HIP_UNSUPPORTED; }
    auto* f = new (std::nothrow) gr4hip_fir64();
    GR4_REQUIRE(f, "out of host memory");
    f->decim = decim;
    int rc   = fir64_upload(f, h_taps, ntaps, false);
    if (rc) { delete f; return rc; }
    *out = f;
    return GR4HIP_OK;
}
int gr4hip_fir64_set_taps(gr4hip_fir64_t* f, const double* h_taps, size_t ntaps) {
    GR4_REQUIRE(f && h_taps && ntaps >= 1, "fir64_set_taps: taps required");
    if (ntaps > 2048) { set_error("fir64: ntaps=%zu is outside the float64 kernel", ntaps); return GR4HIP_UNSUPPORTED; }
    return fir64_upload(f, h_taps, ntaps, true);
}
int gr4hip_fir64_reset(gr4hip_fir64_t* f) {
    GR4_REQUIRE(f, "fir64_reset: null handle");
    f->zero_hist = true;
    return GR4HIP_OK;
}
int gr4hip_fir64_process(gr4hip_fir64_t* f, const double* d_in, size_t n_in, double* d_out, size_t* n_out_p, gr4hip_stream_t stream) {
    GR4_REQUIRE(f, "fir64_process: null handle");
    GR4_REQUIRE(n_in % f->decim == 0, "fir64_process: n_in=%zu is not a multiple of decim=%zu", n_in, f->decim);
    const size_t n_out = n_in / f->decim;
    if (n_out_p) *n_out_p = n_out;
    if (n_in == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in && d_out, "fir64_process: null device pointer");
    hipStream_t st = as_stream(stream);
    if (const int rc = fir64_state_on(f, st)) return rc;
    const int   K = (int)f->taps.size(), D = (int)f->decim;
    if (D == 1 && K > 16 && n_in >= 32768 && (uintptr_t)d_out % 16 == 0 && f->d_trow.ptr) { // matrix-pipe form; writes the next history itself
        const int    Kp = K <= 64 ? 64 : K <= 128 ? 128 : K <= 256 ? 256 : (K + 63) / 64 * 64, NPAD = (kF64Seg + Kp) / 16 * 18;
        const size_t lds = ((size_t)NPAD + Kp + 32) * sizeof(double);
        const dim3   grid((unsigned)ceil_div(ceil_div((long)n_in, (long)kF64Seg), (long)kF64SegPerWg));
        const auto   hp = (const double*)f->d_hist[f->cur].ptr;
        auto         nh = (double*)f->d_hist[f->cur ^ 1].ptr;
        const auto   tr = (const double*)f->d_trow.ptr;
#define GR4_F64_MFMA(KSV)                                                                                                                   \
    do {                                                                                                                                    \
        auto kern = fir64_mfma_kernel<KSV>;                                                                                                 \
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));        \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, d_in, hp, (int)f->hcap, tr, d_out, (long)n_in, nh, (Kp + 16) / 4);               \
    } while (0)
        if (Kp == 64) GR4_F64_MFMA(20); else if (Kp == 128) GR4_F64_MFMA(36); else if (Kp == 256) GR4_F64_MFMA(68); else GR4_F64_MFMA(0);
#undef GR4_F64_MFMA
        GR4_LAUNCH_CHECK();
        f->cur ^= 1;
        return GR4HIP_OK;
    }
    const int   R = D <= 8 ? 4 : D <= 16 ? 2 : 1;
    const size_t lds = ((size_t)K + (size_t)(kF64BS * R - 1) * D + K) * sizeof(double);
    const auto  kern = R == 4 ? fir64_kernel<4> : R == 2 ? fir64_kernel<2> : fir64_kernel<1>;
    if (lds > 64 * 1024) GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const unsigned grid = (unsigned)ceil_div(n_out, (size_t)kF64BS * R);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kF64BS), lds, st, d_in, (const double*)f->d_hist[f->cur].ptr, (int)f->hcap, (const double*)f->d_taps.ptr, K, D, d_out, (long)n_out, (long)n_in);
    GR4_LAUNCH_CHECK();
    hipLaunchKernelGGL(hist64_kernel, dim3((unsigned)ceil_div(f->hcap, (size_t)256)), dim3(256), 0, st, d_in, (const double*)f->d_hist[f->cur].ptr, (double*)f->d_hist[f->cur ^ 1].ptr,
                       (long)n_in, (int)f->hcap);
    GR4_LAUNCH_CHECK();
    f->cur ^= 1;
    return GR4HIP_OK;
}
int gr4hip_fir64_destroy(gr4hip_fir64_t* f) { delete f; return GR4HIP_OK; }

// ---------------------------------------------------------------- iir_filter<double, form>
int gr4hip_iir64_create(gr4hip_iir64_t** out, int form, size_t nsections, const double* h_b, size_t nb, const double* h_a, size_t na) {
    GR4_REQUIRE(out && h_b && h_a && nsections >= 1, "iir64: coefficients required");
    GR4_REQUIRE(form >= 0 && form <= 3, "iir64: unknown form %d", form); // all four forms realise the same transfer function; evaluated as DF-II (like the float kernels)
    GR4_REQUIRE(nb >= 1 && na >= 1, "iir64: empty coefficient rows");
    const size_t ord = std::max(nb, na) - 1;
    if (ord == 0 || ord > 8 || nsections * ord > (size_t)kI64MP) {
        set_error("iir64: %zu sections of order %zu: the float64 kernels hold up to %d state values (4 biquads, or one section of order <= 8)", nsections, ord, kI64MP);
        return GR4HIP_UNSUPPORTED;
    }
    auto* f = new (std::nothrow) gr4hip_iir64();
    GR4_REQUIRE(f, "out of host memory");
    f->d.nsec = (int)nsections;
    f->d.ord  = (int)ord;
    for (size_t c = 0; c < nsections; ++c) {
        const double a0 = h_a[c * na];
        if (a0 == 0.0) { delete f; set_error("iir64: a[0] = 0 in section %zu", c); return GR4HIP_INVALID_ARGUMENT; }
        for (size_t k = 0; k <= ord; ++k) { // normalised by a[0] (FilterTool.hpp Section)
            f->d.b[c][k] = k < nb ? h_b[c * nb + k] / a0 : 0.0;
            f->d.a[c][k] = k < na ? h_a[c * na + k] / a0 : 0.0;
        }
    }
    const int mp = f->mp = (int)(nsections * ord);
    // Phi_L: the zero-input state transition over L samples, column j = the state after L steps from the unit state e_j (host restatement of iir64_step)
    auto step = [&](std::vector<double>& s, double x) {
        for (int c = 0; c < f->d.nsec; ++c) {
            double* sc = s.data() + c * f->d.ord;
            double  w  = x;
            for (int k = 1; k <= f->d.ord; ++k) w -= f->d.a[c][k] * sc[k - 1];
            double yv = f->d.b[c][0] * w;
            for (int k = 1; k <= f->d.ord; ++k) yv += f->d.b[c][k] * sc[k - 1];
            for (int k = f->d.ord - 1; k > 0; --k) sc[k] = sc[k - 1];
            sc[0] = w;
            x     = yv;
        }
    };
    std::vector<long double> PL((size_t)mp * mp);
    for (int j = 0; j < mp; ++j) {
        std::vector<double> s(mp, 0.0);
        s[j] = 1.0;
        for (int i = 0; i < kI64L; ++i) step(s, 0.0);
        for (int i = 0; i < mp; ++i) PL[(size_t)i * mp + j] = s[i];
    }
    auto mul = [mp](const std::vector<long double>& A, const std::vector<long double>& B) {
        std::vector<long double> C((size_t)mp * mp, 0.0L);
        for (int i = 0; i < mp; ++i)
            for (int k = 0; k < mp; ++k)
                for (int j = 0; j < mp; ++j) C[(size_t)i * mp + j] += A[(size_t)i * mp + k] * B[(size_t)k * mp + j];
        return C;
    };
    std::vector<double> pow2((size_t)8 * mp * mp), powl((size_t)kI64BS * mp * mp), phiB((size_t)9 * mp * mp); // phiB: Phi_B, then Phi_B^(G 2^k), k < 8 (pass B's scan)
    {
        std::vector<long double> M = PL;
        for (int k = 0; k < 8; ++k) { // Phi_L^(2^k)
            for (size_t i = 0; i < M.size(); ++i) pow2[(size_t)k * mp * mp + i] = (double)M[i];
            M = mul(M, M);
        }
        for (size_t i = 0; i < M.size(); ++i) phiB[i] = (double)M[i]; // Phi_L^256 = one tile
        {
            std::vector<long double> B = M;
            for (int q = 1; q < kI64G; q <<= 1) B = mul(B, B); // Phi_B^G (G = 16)
            for (int k = 0; k < 8; ++k) {
                for (size_t i = 0; i < B.size(); ++i) phiB[(size_t)(1 + k) * mp * mp + i] = (double)B[i];
                B = mul(B, B);
            }
        }
        std::vector<long double> Q((size_t)mp * mp, 0.0L);
        for (int i = 0; i < mp; ++i) Q[(size_t)i * mp + i] = 1.0L;
        for (int l = 0; l < kI64BS; ++l) { // Phi_L^l
            for (size_t i = 0; i < Q.size(); ++i) powl[(size_t)l * mp * mp + i] = (double)Q[i];
            Q = mul(Q, PL);
        }
    }
    int rc = f->d_phiPow.ensure(pow2.size() * sizeof(double));
    if (!rc) rc = f->d_phiL.ensure(powl.size() * sizeof(double));
    if (!rc) rc = f->d_phiB.ensure(phiB.size() * sizeof(double));
    for (auto& s : f->d_state)
        if (!rc) rc = s.ensure(kI64MP * sizeof(double));
    if (rc) { delete f; return rc; }
    hipError_t e = upload_fresh(f->d_phiPow.ptr, pow2.data(), pow2.size() * sizeof(double));
    if (e == hipSuccess) e = upload_fresh(f->d_phiL.ptr, powl.data(), powl.size() * sizeof(double));
    if (e == hipSuccess) e = upload_fresh(f->d_phiB.ptr, phiB.data(), phiB.size() * sizeof(double));
    if (e != hipSuccess) { delete f; set_error("iir64: upload failed: %s", hipGetErrorString(e)); return GR4HIP_RUNTIME_ERROR; }
    *out = f;
    return GR4HIP_OK;
}
int gr4hip_iir64_reset(gr4hip_iir64_t* f) {
    GR4_REQUIRE(f, "iir64_reset: null handle");
    f->zero_state = true;
    return GR4HIP_OK;
}
int gr4hip_iir64_process(gr4hip_iir64_t* f, const double* d_in, size_t n, double* d_out, gr4hip_stream_t stream) {
    GR4_REQUIRE(f, "iir64_process: null handle");
    if (n == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in && d_out, "iir64_process: null device pointer");
    hipStream_t st    = as_stream(stream);
    if (f->zero_state) { // a pending reset: onto this call's stream, in front of its launches
        for (auto& s : f->d_state) GR4_HIP_TRY(hipMemsetAsync(s.ptr, 0, kI64MP * sizeof(double), st));
        f->zero_state = false;
    }
    const long  tiles = (long)ceil_div(n, (size_t)kI64BS * kI64L);
    const int   mp    = f->mp;
    int         rc    = f->d_P.ensure((size_t)tiles * kI64BS * mp * sizeof(double));
    if (!rc) rc = f->d_Z.ensure((size_t)tiles * mp * sizeof(double));
    if (!rc) rc = f->d_T.ensure((size_t)tiles * mp * sizeof(double));
    if (rc) return rc;
    const double *pp = (const double*)f->d_phiPow.ptr, *pl = (const double*)f->d_phiL.ptr, *pb = (const double*)f->d_phiB.ptr, *s_in = (const double*)f->d_state[f->cur].ptr;
    double *     s_out = (double*)f->d_state[f->cur ^ 1].ptr, *P = (double*)f->d_P.ptr, *Z = (double*)f->d_Z.ptr, *T = (double*)f->d_T.ptr;
#define GR4_IIR64_CASE(NS, OR) \
    if (f->d.nsec == NS && f->d.ord == OR) rc = iir64_run<NS, OR>(f->d, d_in, (long)n, tiles, pp, pl, pb, s_in, s_out, P, Z, T, d_out, st); else
    GR4_IIR64_CASE(1, 1) GR4_IIR64_CASE(2, 1) GR4_IIR64_CASE(3, 1) GR4_IIR64_CASE(4, 1) GR4_IIR64_CASE(5, 1) GR4_IIR64_CASE(6, 1) GR4_IIR64_CASE(7, 1) GR4_IIR64_CASE(8, 1)
    GR4_IIR64_CASE(1, 2) GR4_IIR64_CASE(2, 2) GR4_IIR64_CASE(3, 2) GR4_IIR64_CASE(4, 2) GR4_IIR64_CASE(1, 3) GR4_IIR64_CASE(2, 3) GR4_IIR64_CASE(1, 4) GR4_IIR64_CASE(2, 4)
    GR4_IIR64_CASE(1, 5) GR4_IIR64_CASE(1, 6) GR4_IIR64_CASE(1, 7) GR4_IIR64_CASE(1, 8)
    { set_error("iir64: no kernel for %d sections of order %d", f->d.nsec, f->d.ord); rc = GR4HIP_UNSUPPORTED; }
#undef GR4_IIR64_CASE
    if (rc) return rc;
    f->cur ^= 1;
    return GR4HIP_OK;
}
int gr4hip_iir64_destroy(gr4hip_iir64_t* f) { delete f; return GR4HIP_OK; }

// ---------------------------------------------------------------- FFT<double>
int gr4hip_fft64_create(gr4hip_fft64_t** out, size_t fft_size, int window, int flags) {
    GR4_REQUIRE(out, "fft64: null output handle");
    GR4_REQUIRE(window >= GR4HIP_WIN_NONE && window <= GR4HIP_WIN_KAISER, "fft64: unknown window %d", window);
    if (!is_pow2(fft_size) || fft_size < 2 || fft_size > 8192) {
        set_error("fft64: size %zu is outside the float64 kernel (powers of two 2 .. 8192)", fft_size);
        return GR4HIP_UNSUPPORTED;
    }
    auto* f = new (std::nothrow) gr4hip_fft64();
    GR4_REQUIRE(f, "out of host memory");
    f->N     = fft_size;
    f->log2n = ilog2(fft_size);
    f->flags = flags;
    std::vector<double> tw(fft_size); // W_N^j, j < N/2, interleaved
    for (size_t j = 0; j < fft_size / 2; ++j) {
        const double a = -2.0 * M_PI * (double)j / (double)fft_size;
        tw[2 * j]     = std::cos(a);
        tw[2 * j + 1] = std::sin(a);
    }
    int rc = f->d_tw.ensure(std::max<size_t>(tw.size(), 2) * sizeof(double));
    if (!rc && window != GR4HIP_WIN_NONE && window != GR4HIP_WIN_RECTANGULAR) {
        std::vector<double> w(fft_size);
        rc = make_window64(window, w.data(), fft_size, 1.6);
        if (!rc) rc = f->d_win.ensure(fft_size * sizeof(double));
        if (!rc && upload_fresh(f->d_win.ptr, w.data(), fft_size * sizeof(double)) != hipSuccess) { set_error("fft64: window upload failed"); rc = GR4HIP_RUNTIME_ERROR; }
        f->windowed = true;
    }
    if (!rc && upload_fresh(f->d_tw.ptr, tw.data(), tw.size() * sizeof(double)) != hipSuccess) { set_error("fft64: twiddle upload failed"); rc = GR4HIP_RUNTIME_ERROR; }
    if (rc) { delete f; return rc; }
    *out = f;
    return GR4HIP_OK;
}
int gr4hip_fft64_process(gr4hip_fft64_t* f, const double* d_in, size_t n_frames, double* d_mag, double* d_phase, double* d_re, double* d_im, gr4hip_stream_t stream) {
    GR4_REQUIRE(f, "fft64_process: null handle");
    if (n_frames == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in, "fft64_process: null input");
    hipStream_t  st  = as_stream(stream);
    if (f->N >= 16 && (reinterpret_cast<uintptr_t>(d_in) & 15) == 0) { // half-size complex transform, radix-4 Stockham passes, 8192 / N frames per workgroup
        const size_t lds = (size_t)kF64Pts * sizeof(double2);
        const size_t fpb = kF64Pts / (f->N / 2);
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fft64_r2c_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        Fft64Out o{d_mag, d_phase, d_re, d_im, (f->flags & GR4HIP_FFT_OUTPUT_IN_DB) != 0, (f->flags & GR4HIP_FFT_OUTPUT_IN_DEG) != 0, (f->flags & GR4HIP_FFT_UNWRAP_PHASE) != 0};
        hipLaunchKernelGGL(fft64_r2c_kernel, dim3((unsigned)ceil_div(n_frames, fpb)), dim3(256), lds, st, d_in, f->windowed ? (const double*)f->d_win.ptr : nullptr, (const double2*)f->d_tw.ptr,
                           f->log2n, (long)n_frames, o);
        GR4_LAUNCH_CHECK();
        return GR4HIP_OK;
    }
    const size_t lds = f->N * sizeof(double2);
    if (lds > 64 * 1024) GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fft64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    Fft64Out o{d_mag, d_phase, d_re, d_im, (f->flags & GR4HIP_FFT_OUTPUT_IN_DB) != 0, (f->flags & GR4HIP_FFT_OUTPUT_IN_DEG) != 0, (f->flags & GR4HIP_FFT_UNWRAP_PHASE) != 0};
    hipLaunchKernelGGL(fft64_kernel, dim3((unsigned)n_frames), dim3(256), lds, st, d_in, f->windowed ? (const double*)f->d_win.ptr : nullptr, (const double2*)f->d_tw.ptr, f->log2n,
                       (long)n_frames, o);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}
int gr4hip_fft64_destroy(gr4hip_fft64_t* f) { delete f; return GR4HIP_OK; }

// ---------------------------------------------------------------- Rotator<complex<double>>
int gr4hip_rotator64_create(gr4hip_rotator64_t** out, double phase_increment, double initial_phase) {
    GR4_REQUIRE(out, "rotator64: null output handle");
    auto* r = new (std::nothrow) gr4hip_rotator64();
    GR4_REQUIRE(r, "out of host memory");
    r->inc = phase_increment;
    r->phase = r->initial = initial_phase;
    *out = r;
    return GR4HIP_OK;
}
int gr4hip_rotator64_reset(gr4hip_rotator64_t* r, double initial_phase) {
    GR4_REQUIRE(r, "rotator64_reset: null handle");
    r->phase = r->initial = initial_phase;
    return GR4HIP_OK;
}
int gr4hip_rotator64_process(gr4hip_rotator64_t* r, const void* d_in_c64, void* d_out_c64, size_t n, gr4hip_stream_t stream) {
    GR4_REQUIRE(r, "rotator64_process: null handle");
    if (n == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in_c64 && d_out_c64, "rotator64_process: null device pointer");
    const unsigned grid = (unsigned)std::min<size_t>(ceil_div(n, (size_t)256), 1u << 16);
    hipLaunchKernelGGL(rotator64_kernel, dim3(grid), dim3(256), 0, as_stream(stream), (const double2*)d_in_c64, (double2*)d_out_c64, (long)n, r->phase, r->inc);
    GR4_LAUNCH_CHECK();
    const double two_pi = 6.283185307179586476925286766559;
    double       ph     = std::fma((double)n, r->inc, r->phase); // the accumulated phase after n samples, kept in [0, 2 pi) like the block's wrap (Rotator.hpp:55-60)
    ph -= two_pi * std::floor(ph / two_pi);
    r->phase = ph;
    return GR4HIP_OK;
}
int gr4hip_rotator64_phase(gr4hip_rotator64_t* r, double* phase) {
    GR4_REQUIRE(r && phase, "rotator64_phase: null argument");
    *phase = r->phase;
    return GR4HIP_OK;
}
int gr4hip_rotator64_destroy(gr4hip_rotator64_t* r) { delete r; return GR4HIP_OK; }

} // extern "C"
