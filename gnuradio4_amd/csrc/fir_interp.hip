// fir_interp.hip -- polyphase interpolating FIR for gfx950 (BASELINE.json north_star: "decimating / interpolating FIR").
//
// The reference has no interpolating filter block -- only the rate machinery a block would declare, Resampling<1, L>
// (core/include/gnuradio-4.0/annotated.hpp:121-128, chunk bookkeeping Block.hpp:1576-1636).  Definition: zero-stuff the
// input by L, run fir_filter's sum (blocks/filter/.../time_domain_filter.hpp:44-47) at the output rate, gain L:
//     u[n] = x[n / L] if n % L == 0 else 0,       y[n] = L sum_k b[k] u[n - k]
// Only every L-th product is non-zero, so output n = m L + p is branch p of a polyphase bank at the INPUT rate:
//     y[m L + p] = sum_q (L b[q L + p]) x[m - q],   q < Kp = ceil(K / L)
// i.e. L short FIRs over the same input window.  A lane owns R consecutive input positions and all L phases of them: R L accumulators, one
// sliding register window of the staged input (one LDS read per tap step), the L branch taps of step q are wave-uniform (scalar loads), and the
// R L outputs of a lane are contiguous in memory.  Bound: FP32 FMA rate at long filters (2 K / L flop per output), HBM at short ones
// ((4 + 4 L) B per real input sample).  History: the last Kp - 1 input samples (capacity max(32, bit_ceil(Kp)) like HistoryBuffer, :36-42).
#include "common.hpp"

#include <algorithm>

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
    DeviceBuffer       d_taps, d_hist[2];
    int                cur = 0;
};

static size_t ip_bit_ceil(size_t v) { size_t p = 1; while (p < v) p <<= 1; return p; }

static int ip_upload(gr4hip_fir_interp* f) {
    f->Kp = ceil_div(f->ntaps, f->L);
    std::vector<float> t(f->Kp * f->L, 0.f);
    for (size_t k = 0; k < f->ntaps; ++k) t[(k / f->L) * f->L + (k % f->L)] = (float)f->L * f->taps[k]; // branch p = k % L, step q = k / L, gain L
    int rc = f->d_taps.ensure(t.size() * sizeof(float));
    if (rc) return rc;
    GR4_HIP_TRY(hipMemcpy(f->d_taps.ptr, t.data(), t.size() * sizeof(float), hipMemcpyHostToDevice));
    return GR4HIP_OK;
}
static int ip_alloc_hist(gr4hip_fir_interp* f) {
    const size_t bytes = f->hcap * f->S * sizeof(float);
    for (int k = 0; k < 2; ++k) {
        int rc = f->d_hist[k].ensure(bytes);
        if (rc) return rc;
        GR4_HIP_TRY(hipMemset(f->d_hist[k].ptr, 0, bytes));
    }
    f->cur = 0;
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
    int         rc = f->S == 1 ? ip_dispatch<1>(f, d_in, d_out, (long)n_in, st) : ip_dispatch<2>(f, d_in, d_out, (long)n_in, st);
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
