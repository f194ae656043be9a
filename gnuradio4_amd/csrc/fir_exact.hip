// fir_exact.hip -- the second evaluation of the matrix-pipe FIR / decimator kernels, as ONE kernel behind all of them:  y[o] = sum_k b[k] x[D o - k]  on the FP64 matrix
// pipe (v_mfma_f64_16x16x4_f64: every float32 x float32 product exact in float64, the sums in float64, ONE rounding to float32 at the end).
//
// Why it exists.  The split-product kernels (fir_f16.hip, fir_decim_f16.hip; the fused chain of chain_fused.hip) carry an error that is relative to the PRODUCTS -- like the
// reference's own float32 sum (time_domain_filter.hpp:44-47), but a few times larger -- so it shows against the OUTPUT only when the filter removes nearly everything it is
// given.  Those kernels judge every segment themselves (output power against input power) and mark the ones they cannot vouch for; the marked segments are evaluated again
// HERE, with an error far below the reference's float32 arithmetic whatever the signal (2^-53 of the partial sums instead of 2^-24): the parity contract's second clause
// (include/gr4hip.h, "PARITY CONTRACT") holds with a factor of ONE and a wide margin, with no per-shape exceptions.  Until round 5 every kernel family carried its own second
// evaluation inside its main kernel (three-term f16 products, float32 matrix-pipe sums): 110 live registers spilled around them, and their float32 accumulation measured
// 1 .. 3.3 x the reference's error depending on the shape.
//
// One kernel for every family because the contraction is the same band form for all of them:
//
//     D[j][c] = sum_u A[j][u] B[u][c],   A[j][u] = b[Hb + D j - u],   B[u][c] = staged[D (256 t + 16 c) + u],   u < Kw = Hb + 15 D + 1
//
// -- a tile is 16 consecutive outputs (rows j) of 16 columns c that lie 16 outputs apart: ONE tile row = 256 consecutive outputs, its input window 256 D + Kw samples.  The
// samples are staged as raw float32 (LDS, 4 floats of padding per 16 D: the 64 lanes of a B read hit 64 different banks for every D), the taps as a zero-padded float32 array;
// both operands are converted to float64 on the way into the matrix pipe (one v_cvt_f64_f32 each per 64-cycle MFMA).  A workgroup's unit is 1024 outputs (float: four tile
// rows, one per wave) or 512 complex outputs (real taps never mix the components: they are staged as two real streams, wave = component x tile row).
//
// Which outputs: `flags` holds one byte per segment of the caller's main kernel (2^seg_shift outputs): non-zero = evaluate again (the main kernels write 1: outlier spread,
// 2: a non-finite sample, 3: rejected by the guard); null = every output (the fused chain's guard, gated by the launch's verdict word `gate`).  A unit none of whose segments is
// marked costs its workgroup a few byte loads.  A unit whose staged window holds a non-finite sample is evaluated as plain float32 sums in the reference's order instead, one
// output at a time (0 x Inf in the zero part of the tap operand would spread the NaN to outputs whose window does not hold the sample; the classes +Inf / -Inf / NaN and their
// reach -- exactly ntaps outputs -- are the reference's).
//
// Rate when EVERY segment is marked (measured, profiles/r05_rejected_stream_rates.txt; main kernel + this one): float 256 taps 264 Gsamples/s (518 unmarked), complex 143
// (178), decimate by 8 with 1024 taps 417 G input samples/s (749).  That is the price of a stream that is all rejection; ordinary streams never come here.
#include "common.hpp"
#include "ewise.hpp"
#include "fir_exact.hpp"
#include "fir_f16_common.hpp" // (hf_wave_sum)

#include <algorithm>

namespace gr4 {

using f64x4_e = __attribute__((ext_vector_type(4))) double;

struct FirExactArgs {
    const float*         x;          // the span: n_in samples per channel (complex: interleaved {re, im}), channel ch at x + ch * in_stride floats
    const float*         hist;       // the Kh samples in front of x (channel ch at hist + ch * Kh samples); older samples read as zero
    const float*         taps;       // float32 taps, channel ch at taps + ch * taps_stride
    float*               y;          // n_out outputs per channel, channel ch at y + ch * out_stride floats
    const unsigned char* flags;      // one byte per 2^seg_shift outputs (channel ch at flags + ch * flags_stride), or null: every output
    const unsigned*      gate;       // optional: the kernel does nothing unless *gate != 0
    long                 n_in, n_out, in_stride, out_stride, taps_stride, flags_stride;
    int                  Kh, ntaps, D, cplx, seg_shift;
    int                  Hb, Kw;     // samples in front of a tile's first output that its window holds (>= ntaps - 1), window length Hb + 15 D + 1 (a multiple of 16)
    unsigned             div16D;     // ceil(2^32 / (16 D)): i / (16 D) = (i * div16D) >> 32 for the indices that occur
    long                 n_units;
    EwiseHook            pre, post;  // HOOK: the main kernel's load / store programs (the stored history holds prologue OUTPUTS already: only samples of x are mapped)
};

constexpr int kExUnit = 1024; // outputs (floats) per unit: 4 tile rows; complex: 512 outputs x 2 components

template <bool HOOK>
__global__ __launch_bounds__(256, 1) void fir_exact_kernel(FirExactArgs a) {
    if (a.gate != nullptr && __builtin_nontemporal_load(a.gate) == 0u) return;
    extern __shared__ __attribute__((aligned(16))) float ex_smem[];
    __shared__ int bad;
    const int ch = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4;
    const int NC = a.cplx ? 2 : 1, UO = kExUnit / NC, D = a.D, Hb = a.Hb, Kw = a.Kw; // components, outputs per component and unit
    const float*         x     = a.x + (long)ch * a.in_stride;
    const float*         hist  = a.hist + (long)ch * a.Kh * NC;
    const float*         taps  = a.taps + (long)ch * a.taps_stride;
    float*               y     = a.y + (long)ch * a.out_stride;
    const unsigned char* flags = a.flags ? a.flags + (long)ch * a.flags_stride : nullptr;
    const int L  = D * (UO - 1) + Hb + 1;                                  // staged samples per component
    auto      ph = [&](int i) { return i + 4 * (int)(((unsigned long long)(unsigned)i * a.div16D) >> 32); }; // LDS position of staged sample i
    const int Lp = ph(L - 1) + 4;
    float*    tz = ex_smem;                    // tz[15 D + k] = b[k], zeros either side: k = Hb + D j - u runs over -15 D .. Kw - 1
    float*    sg = ex_smem + ((Kw + 15 * D + 3) & ~3); // [NC][Lp]
    auto xs = [&](long i, int c) -> float { // sample i of the stream, component c: history in front of x, zeros before that and past the end
        if (i >= 0) {
            if (i >= a.n_in) return 0.f;
            if constexpr (HOOK) {
                if (a.pre.n_ops > 0) {
                    if (NC == 1) return ewise_hook1<float>(x[i], a.pre, i);
                    const float2 q = ewise_hook1<float2>(make_float2(x[2 * i], x[2 * i + 1]), a.pre, i);
                    return c ? q.y : q.x;
                }
            }
            return x[i * NC + c];
        }
        return i >= -(long)a.Kh ? hist[((long)a.Kh + i) * NC + c] : 0.f;
    };
    auto put = [&](long o, int c, float v, float v_other) { // output o, component c (complex: v_other = the other component, for a hooked store)
        if constexpr (HOOK) {
            if (a.post.n_ops > 0) {
                if (NC == 1) v = ewise_hook1<float>(v, a.post, o);
                else { const float2 q = ewise_hook1<float2>(c ? make_float2(v_other, v) : make_float2(v, v_other), a.post, o); v = c ? q.y : q.x; }
            }
        }
        y[o * NC + c] = v;
    };
    if (flags != nullptr) { // an ordinary stream marks nothing: every lane looks at its share of the workgroup's flag bytes at once and the workgroup leaves (a lane that walks
                            // them one dependent load after the other cost configs[3] 50 us of its 520)
        const long nseg = ((a.n_out - 1) >> a.seg_shift) + 1;
        int        any  = 0;
        for (long u = (long)blockIdx.x + (long)tid * gridDim.x; u < a.n_units; u += 256L * gridDim.x) {
            const long s0 = (u * UO) >> a.seg_shift, s1 = ((u + 1) * UO - 1) >> a.seg_shift;
            for (long sgi = s0; sgi <= s1 && sgi < nseg; ++sgi) any |= flags[sgi];
        }
        if (!__syncthreads_or(any)) return;
    }
    for (int i = tid; i < Kw + 15 * D; i += 256) {
        const int k = i - 15 * D;
        tz[i]       = (k >= 0 && k < a.ntaps) ? taps[k] : 0.f;
    }
    const int comp = a.cplx ? (wave & 1) : 0, tr = a.cplx ? (wave >> 1) : wave; // this wave's component and tile row
    for (long u = blockIdx.x; u < a.n_units; u += gridDim.x) {
        const long ou = u * UO; // first output of the unit
        bool       any = flags == nullptr;
        if (!any) {
            const long last = (ou + UO - 1 < a.n_out ? ou + UO - 1 : a.n_out - 1) >> a.seg_shift;
            for (long s = ou >> a.seg_shift; s <= last; ++s) any = any || flags[s] != 0; // (uniform: every lane reads the same bytes)
        }
        if (!any) continue;
        __syncthreads(); // (the previous unit's readers are done with the staged samples)
        if (tid == 0) bad = 0;
        __syncthreads();
        const long s0 = (long)D * ou - Hb;
        int        nf = 0;
        for (int i = tid; i < L; i += 256) {
            const int p = ph(i);
            for (int c = 0; c < NC; ++c) {
                const float v = xs(s0 + i, c);
                nf |= (int)!(__builtin_fabsf(v) <= 3.4028234663852886e38f);
                sg[c * Lp + p] = v;
            }
        }
        if (nf) bad = 1;
        __syncthreads();
        auto marked = [&](long o) -> bool { return o < a.n_out && (flags == nullptr || flags[o >> a.seg_shift] != 0); };
        if (bad) { // a non-finite sample in the window: plain float32 sums in the reference's order
            for (int r = tid; r < UO; r += 256) {
                const long o = ou + r;
                if (!marked(o)) continue;
                float     acc[2] = {0.f, 0.f};
                const int b0     = D * r + Hb;
                for (int c = 0; c < NC; ++c)
                    for (int k = 0; k < a.ntaps; ++k) acc[c] = fmaf(tz[15 * D + k], sg[c * Lp + ph(b0 - k)], acc[c]);
                for (int c = 0; c < NC; ++c) put(o, c, acc[c], acc[c ^ 1]);
            }
            continue;
        }
        const bool swap = HOOK && NC == 2 && a.post.n_ops > 0; // a complex store program wants both components of an output: they sit in the neighbouring wave, exchanged through LDS
        if (!swap && (long)ou + 256L * tr >= a.n_out) continue; // (whole waves; no barrier below)
        // this wave's tile row: B[u][c] = staged[D (256 tr + 16 c) + u], A[j][u] = tz[15 D + Hb + D j - u]
        const float* sb = sg + comp * Lp + D * (256 * tr + 16 * col) + 4 * (16 * tr + col) + kq;
        const float* ta = tz + 15 * D + Hb + D * col - kq;
        f64x4_e      acc = {0., 0., 0., 0.}, acc1 = {0., 0., 0., 0.}; // two accumulators over alternating K-steps: a single chain of dependent MFMAs leaves the pipe idle for its latency
        const int    nk = Kw / 4, blk = 4 * D; // K-steps of 4; a pad of 4 floats every 16 D samples = every 4 D steps
        for (int k0 = 0, pad = 0; k0 < nk; k0 += blk, pad += 4) {
            const int kend = k0 + blk < nk ? k0 + blk : nk;
            for (int k4 = k0; k4 < kend; k4 += 4) { // (Kw is a multiple of 16: whole groups of four K-steps)
                float av[4], bv[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) { av[q] = ta[-4 * (k4 + q)]; bv[q] = sb[4 * (k4 + q) + pad]; }
                acc  = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[0], (double)bv[0], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[1], (double)bv[1], acc1, 0, 0, 0);
                acc  = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[2], (double)bv[2], acc, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64((double)av[3], (double)bv[3], acc1, 0, 0, 0);
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += acc1[r];
        // D[row = kq + 4 r][col]: output ou + 256 tr + 16 col + kq + 4 r
        float other[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (HOOK) {
            if (swap) {
                float* ox = sg + NC * Lp; // [4 waves][4][64]
#pragma unroll
                for (int r = 0; r < 4; ++r) ox[(wave * 4 + r) * 64 + lane] = (float)acc[r];
                __syncthreads();
#pragma unroll
                for (int r = 0; r < 4; ++r) other[r] = ox[((wave ^ 1) * 4 + r) * 64 + lane];
            }
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const long o = ou + 256L * tr + 16 * col + kq + 4 * r;
            if (marked(o)) put(o, comp, (float)acc[r], other[r]);
        }
    }
}

// Hb, Kw of the band form for (ntaps, D); false when the staged unit does not fit the CU's 160 KiB of LDS
bool fir_exact_shape(int ntaps, int D, int cplx, int* Hb, int* Kw, size_t* lds_bytes) {
    if (ntaps < 1 || D < 1 || D > 64) return false;
    const int kw = (ntaps + 15 * D + 15) & ~15, hb = kw - 15 * D - 1, NC = cplx ? 2 : 1, UO = kExUnit / NC;
    const long L  = (long)D * (UO - 1) + hb + 1, Lp = (L - 1) + 4 * ((L - 1) / (16 * D)) + 4;
    const size_t bytes = (size_t)(((kw + 15 * D + 3) & ~3) + NC * Lp + (cplx ? 1024 : 0)) * sizeof(float); // (+ the hooked complex store's exchange rows)
    if (Hb) *Hb = hb;
    if (Kw) *Kw = kw;
    if (lds_bytes) *lds_bytes = bytes;
    return bytes <= 158 * 1024;
}

// evaluates the marked outputs (flags == nullptr: all of them) of  y[o] = sum_k b[k] x[D o - k],  o < n_out.  Sizes in samples (float, or complex when cplx); strides in floats.
int fir_exact_launch(const float* x, long n_in, const float* hist, int Kh, const float* d_taps, int ntaps, int D, int cplx, float* y, long n_out, const unsigned char* flags, int seg_shift,
                     const unsigned* gate, hipStream_t st, unsigned nch, long in_stride, long out_stride, long taps_stride, long flags_stride, const EwiseHook* pre, const EwiseHook* post) {
    FirExactArgs a{};
    size_t       lds = 0;
    if (!fir_exact_shape(ntaps, D, cplx, &a.Hb, &a.Kw, &lds)) return GR4HIP_UNSUPPORTED;
    if (n_out <= 0) return GR4HIP_OK;
    static PerDevice pd;
    bool             first = false;
    int              dev = 0, n_cu = pd.current(&first, &dev);
    if (first) {
        if (n_cu == 0) { set_error("fir_exact: no device"); return GR4HIP_NO_DEVICE; }
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fir_exact_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fir_exact_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
        n_cu = -n_cu;
        pd.done(dev, n_cu);
    }
    a.x = x; a.hist = hist; a.taps = d_taps; a.y = y; a.flags = flags; a.gate = gate;
    a.n_in = n_in; a.n_out = n_out; a.in_stride = in_stride; a.out_stride = out_stride; a.taps_stride = taps_stride; a.flags_stride = flags_stride;
    a.Kh = Kh; a.ntaps = ntaps; a.D = D; a.cplx = cplx; a.seg_shift = seg_shift;
    a.div16D  = (unsigned)(((1ull << 32) + 16ull * D - 1) / (16ull * D));
    a.n_units = ceil_div(n_out, (long)(kExUnit / (cplx ? 2 : 1)));
    const unsigned gx = (unsigned)std::min<long>(a.n_units, std::max<long>(1, (long)n_cu * 4 / (long)nch));
    const bool hooked = (pre && pre->n_ops > 0) || (post && post->n_ops > 0);
    if (hooked) {
        if (pre) a.pre = *pre;
        if (post) a.post = *post;
        hipLaunchKernelGGL(fir_exact_kernel<true>, dim3(gx, nch), dim3(256), lds, st, a);
    } else hipLaunchKernelGGL(fir_exact_kernel<false>, dim3(gx, nch), dim3(256), lds, st, a);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

// ---- the verdict for kernels that do not judge themselves (the float32 matrix-pipe forms, the three-term bf16 direct forms): one byte per 2^seg_shift outputs, non-zero where the
// outputs' power (of the segment's quietest quarter) is below gthr x the power of the D x as many samples in front of them.  One pass over x and y; non-finite power either
// side leaves the segment unmarked (the main kernel's classes stand).
__global__ __launch_bounds__(256) void fir_judge_kernel(const float* __restrict__ x, long nxf, const float* __restrict__ y, long nyf, int D, int segf, float gthr, float gthr_all, unsigned char* __restrict__ flags, long nseg) {
    __shared__ float red[8];
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63, q = segf / 4; // a wave's quarter: q output floats, q D input floats
    for (long sgi = blockIdx.x; sgi < nseg; sgi += gridDim.x) {
        const long y0 = sgi * segf + (long)wave * q, x0 = y0 * D;
        float      px = 0.f, py = 0.f;
        for (long i = y0 + lane; i < y0 + q && i < nyf; i += 64) { const float v = y[i]; py = fmaf(v, v, py); }
        for (long i = x0 + lane; i < x0 + (long)q * D && i < nxf; i += 64) { const float v = x[i]; px = fmaf(v, v, px); }
        px = hf_wave_sum(px);
        py = hf_wave_sum(py);
        __syncthreads();
        if (lane == 0) { red[wave] = px; red[4 + wave] = py; }
        __syncthreads();
        if (tid == 0) {
            const float pxs = (red[0] + red[1]) + (red[2] + red[3]);
            float       pym = red[4], pya = 0.f;
            bool        fin = true;
            for (int w = 0; w < 4; ++w) {
                const bool present = sgi * segf + (long)w * q < nyf; // (the stream's last segment: quarters past its end do not vote)
                if (!present) continue;
                fin = fin && red[4 + w] < 3.0e38f;
                pym = fminf(pym, red[4 + w]);
                pya += red[4 + w];
            }
            flags[sgi] = (fin && pxs < 3.0e38f && (pym * 4.f * (float)D < gthr * pxs || pya * (float)D < gthr_all * pxs)) ? 3 : 0; // (pya over the quarters present: the stream's last segment is judged a little early, never late)
        }
    }
}
int fir_judge_launch(const float* x, long n_in, const float* y, long n_out, int D, int cplx, int seg_shift, float gthr, unsigned char* flags, hipStream_t st, float gthr_all) {
    const int  NC = cplx ? 2 : 1, segf = NC << seg_shift;
    const long nseg = ceil_div(n_out, 1L << seg_shift);
    if (nseg <= 0) return GR4HIP_OK;
    hipLaunchKernelGGL(fir_judge_kernel, dim3((unsigned)std::min<long>(nseg, 256 * 16)), dim3(256), 0, st, x, n_in * NC, y, n_out * NC, D, segf, gthr, gthr_all, flags, nseg);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

} // namespace gr4
