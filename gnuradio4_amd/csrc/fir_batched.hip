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

namespace gr4 {

using f32x4 = __attribute__((ext_vector_type(4))) float;

constexpr int kSeg = 4096; // output samples per workgroup (4 waves x 4 tiles x 256)

template <int KS> // K-steps of 4: Kp = 4 KS - 16
__global__ __launch_bounds__(256) void fir_mfma_kernel(const float* __restrict__ x, long in_stride, const float* __restrict__ hist, const float* __restrict__ afrag,
                                                        float* __restrict__ y, long out_stride, long n) {
    constexpr int Kp   = 4 * KS - 16;
    constexpr int NPAD = (kSeg + Kp) / 16 * 18; // two pad floats per 16 samples: block stride 18 = 2 mod 32 banks, so the 32 lanes (16 blocks x 2 K
                                                // offsets) of a ds_read_b32 group hit 32 different banks (stride 17 puts two of them on one)
    __shared__ float xs[NPAD];
    const int  c    = blockIdx.y;
    const long seg0 = (long)blockIdx.x * kSeg;
    const int  tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* xc = x + (long)c * in_stride;
    const float* hc = hist + (long)c * Kp;

    for (int s = tid; s < kSeg + Kp; s += 256) { // staged index s <-> input index seg0 - Kp + s; one pad float per 16 samples
        const long i = seg0 - Kp + s;
        const float v = i >= 0 ? (i < n ? xc[i] : 0.f) : hc[Kp + i];
        xs[s + 2 * (s >> 4)] = v;
    }
    float a[KS]; // A fragments: lane l holds A[j = l & 15][u = 4 ks + (l >> 4)] = b[Kp + j - u]
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a[ks] = afrag[((long)c * KS + ks) * 64 + lane];
    __syncthreads();

    const int col = lane & 15, kq = lane >> 4;
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
        float*     yc = y + (long)c * out_stride;
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
// phase rows while it is staged (every input read from HBM once); the A fragments of one phase at a time are in registers.
template <int KS, int TPW> // K-steps per phase (Kp = 4 KS - 16), 256-output tiles per wave
__global__ __launch_bounds__(256) void fir_mfma_decim_kernel(const float* __restrict__ x, const float* __restrict__ hist /*[Kp D] samples in front of x*/,
                                                              const float* __restrict__ afrag /*[D][KS][64]*/, float* __restrict__ y, long n_out, int D) {
    constexpr int Kp  = 4 * KS - 16;
    constexpr int SEG = 1024 * TPW;                  // outputs per workgroup
    constexpr int ROW = (SEG + Kp) / 16 * 17 + 1;    // padded phase row, one pad per 16 (odd length: the D rows start on different banks).  The
                                                     // conflict-free stride 18 of fir_mfma_kernel costs the fourth workgroup per CU at D = 8 here (-4 %)
    extern __shared__ float xs[];                    // [D][ROW]
    const long seg0 = (long)blockIdx.x * SEG;        // first output of this workgroup
    const int  tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long n_in = n_out * D, H = (long)Kp * D;
    using f32x4 = __attribute__((ext_vector_type(4))) float;
    const int col = lane & 15, kq = lane >> 4;
    // A fragments of phase 0 are requested before the staging loop and every following phase one phase ahead: their L2 latency hides
    // under the staging / the previous phase's MFMAs (A does not depend on the wave: four waves share the lines in L1)
    float a0[KS], a1[KS];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) a0[ks] = afrag[(long)ks * 64 + lane];

    // stage: phase row p, position m' <-> input index (seg0 - Kp + m') D - p.  Consecutive lanes take consecutive input samples; the
    // (m', p) pair of a lane's next sample (256 further) follows from the previous one without a division.
    const long i0  = (seg0 - Kp) * D - (D - 1);      // lowest input index needed (m' = 0, p = D - 1)
    const int  cnt = (SEG + Kp) * D;
    {
        const int q256 = 256 / D, r256 = 256 % D;
        int       e  = tid - (D - 1);                // = m' D - p for s = tid
        int       mp = (e + D - 1) / D, pp = mp * D - e;
        constexpr int U = 8; // loads in flight per lane (a load-per-iteration loop pays the memory latency cnt / 256 times)
        for (int s0 = tid; s0 < cnt; s0 += 256 * U) {
            float v[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const long i = i0 + s0 + 256 * u;
                v[u] = s0 + 256 * u < cnt ? (i >= 0 ? (i < n_in ? x[i] : 0.f) : (i >= -H ? hist[H + i] : 0.f)) : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                if (s0 + 256 * u < cnt && mp < SEG + Kp) xs[pp * ROW + mp + (mp >> 4)] = v[u];
                mp += q256;
                pp -= r256;
                if (pp < 0) { pp += D; mp += 1; }
            }
        }
    }
    __syncthreads();

    f32x4 acc[TPW]; // one 16-block x 16-output tile (256 outputs) each; two tiles per wave hide the dependent-MFMA latency
#pragma unroll
    for (int t = 0; t < TPW; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    const float* pb[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) pb[t] = xs + kq + 17 * (16 * (wave * TPW + t) + col);
    auto phase = [&](const float (&a)[KS], int p) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const int off = 4 * ks + (ks >> 2);
#pragma unroll
            for (int t = 0; t < TPW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ks], pb[t][p * ROW + off], acc[t], 0, 0, 0);
        }
    };
    for (int p = 0; p < D; p += 2) {
        const int pn = p + 1 < D ? p + 1 : p; // (odd D: the last prefetch re-reads a valid phase and is not used)
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) a1[ks] = afrag[((long)pn * KS + ks) * 64 + lane];
        phase(a0, p);
        if (p + 1 < D) {
            const int pm = p + 2 < D ? p + 2 : p;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) a0[ks] = afrag[((long)pm * KS + ks) * 64 + lane];
            phase(a1, p + 1);
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
int fir_mfma_launch(int KS, const float* x, long in_stride, const float* hist, const float* afrag, float* y, long out_stride, long n, unsigned nch, hipStream_t st) {
    const dim3 grid((unsigned)ceil_div(n, (long)kSeg), nch);
    switch (KS) {
    case 20: hipLaunchKernelGGL(fir_mfma_kernel<20>, grid, dim3(256), 0, st, x, in_stride, hist, afrag, y, out_stride, n); break;
    case 36: hipLaunchKernelGGL(fir_mfma_kernel<36>, grid, dim3(256), 0, st, x, in_stride, hist, afrag, y, out_stride, n); break;
    default: hipLaunchKernelGGL(fir_mfma_kernel<68>, grid, dim3(256), 0, st, x, in_stride, hist, afrag, y, out_stride, n); break;
    }
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

// polyphase A fragments [D][KS][64]: phase p holds b[qD + p]
void fir_mfma_make_afrag_decim(const float* taps, size_t ntaps, size_t D, int* Kp_out, int* KS_out, std::vector<float>* af_out) {
    const size_t Q  = ceil_div(ntaps, D);
    const int    Kp = Q <= 64 ? 64 : Q <= 128 ? 128 : 256, KS = (Kp + 16) / 4;
    af_out->assign(D * KS * 64, 0.f);
    for (size_t p = 0; p < D; ++p)
        for (int ks = 0; ks < KS; ++ks)
            for (int l = 0; l < 64; ++l) {
                const int q = Kp + (l & 15) - (4 * ks + (l >> 4)); // phase-tap index
                if (q >= 0 && (size_t)q * D + p < ntaps) (*af_out)[(p * KS + ks) * 64 + l] = taps[(size_t)q * D + p];
            }
    *Kp_out = Kp;
    *KS_out = KS;
}

// y[m] = sum_k b[k] x[mD - k], m < n_out; hist = the Kp D samples in front of x; y 16-byte aligned.  Returns GR4HIP_UNSUPPORTED when the
// de-interleaved segment does not fit the LDS.
int fir_mfma_decim_launch(int KS, int D, const float* x, const float* hist, const float* afrag, float* y, long n_out, hipStream_t st) {
    const int Kp = 4 * KS - 16;
    for (int tpw : {1, 2}) { // smaller segments first: 39 KB of LDS at D = 8 -> four workgroups per CU cover the staging and A-fragment latencies
        const int    seg = 1024 * tpw, row = (seg + Kp) / 16 * 17 + 1;
        const size_t lds = (size_t)D * row * sizeof(float);
        if (lds > 76 * 1024) continue; // two workgroups per CU
        const dim3 grid((unsigned)ceil_div(n_out, (long)seg));
#define GR4_DECIM_CASE(KSV, TPWV)                                                                                                         \
    do {                                                                                                                                  \
        auto kern = fir_mfma_decim_kernel<KSV, TPWV>;                                                                                     \
        if (lds > 48 * 1024) GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
        hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, x, hist, afrag, y, n_out, D);                                                  \
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

struct gr4hip_fir_batched {
    size_t       nch = 0, ntaps = 0;
    int          KS = 0, Kp = 0;
    DeviceBuffer d_afrag, d_hist[2];
    int          cur = 0;
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
    if (!rc) { hipError_t e = hipMemcpy(f->d_afrag.ptr, af.data(), af.size() * sizeof(float), hipMemcpyHostToDevice); if (e != hipSuccess) { set_error("fir_batched: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
    for (int k = 0; k < 2 && !rc; ++k) rc = f->d_hist[k].ensure(nchannels * f->Kp * sizeof(float));
    if (!rc) rc = gr4hip_fir_batched_reset(f);
    if (rc) { delete f; return rc; }
    *out = f;
    return GR4HIP_OK;
}

int gr4hip_fir_batched_reset(gr4hip_fir_batched_t* f) {
    GR4_REQUIRE(f, "fir_batched_reset: null handle");
    for (int k = 0; k < 2; ++k) GR4_HIP_TRY(hipMemset(f->d_hist[k].ptr, 0, f->nch * f->Kp * sizeof(float)));
    f->cur = 0;
    return GR4HIP_OK;
}

int gr4hip_fir_batched_process(gr4hip_fir_batched_t* f, const float* d_in, size_t in_stride, size_t n, float* d_out, size_t out_stride, gr4hip_stream_t stream) {
    GR4_REQUIRE(f, "fir_batched_process: null handle");
    if (n == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in && d_out && in_stride >= n && out_stride >= n, "fir_batched_process: null pointer or stride shorter than n");
    GR4_REQUIRE(((uintptr_t)d_out % 16 == 0) && (out_stride % 4 == 0), "fir_batched_process: output must be 16-byte aligned with a stride multiple of 4");
    hipStream_t st = as_stream(stream);
    const float *hist = (const float*)f->d_hist[f->cur].ptr, *af = (const float*)f->d_afrag.ptr;
    int          rc   = fir_mfma_launch(f->KS, d_in, (long)in_stride, hist, af, d_out, (long)out_stride, (long)n, (unsigned)f->nch, st);
    if (rc) return rc;
    hipLaunchKernelGGL(fir_batched_hist_kernel, dim3((unsigned)ceil_div(f->Kp, 64), (unsigned)f->nch), dim3(64), 0, st, d_in, (long)in_stride, hist,
                       (float*)f->d_hist[f->cur ^ 1].ptr, (long)n, f->Kp);
    GR4_LAUNCH_CHECK();
    f->cur ^= 1;
    return GR4HIP_OK;
}

int gr4hip_fir_batched_destroy(gr4hip_fir_batched_t* f) { delete f; return GR4HIP_OK; }

} // extern "C"
