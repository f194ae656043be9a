// fft.hip -- generic power-of-two streaming FFT block kernels for gfx950 (LDS-resident Stockham, radix 8/4/2).
//
// Replaces gr::blocks::fft::FFT<T>::processBulk (blocks/fourier/.../fft.hpp:147-171): window -> forward DFT
// (algorithm/.../fourier/fft.hpp:113-153) -> magnitude / phase / Re / Im (fft_common.hpp:20-123), many frames per
// launch instead of one frame per work() call.  One workgroup owns whole frames: the frame is read from HBM once
// (window fused into the load), all passes run in LDS, and every requested output is written once, coalesced.
// The fused FIR->FFT->mag2 headline kernels live in chain.hip; this file is the any-size / any-window path.
#include "common.hpp"

#include <cfloat>
#include <cmath>

namespace gr4 {

__device__ __forceinline__ float2 cmulf(float2 a, float2 b) { return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)); }
__device__ __forceinline__ float2 caddf(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csubf(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); } // * (-i)

__device__ __forceinline__ void fft2(float2& a, float2& b) {
    const float2 t = a;
    a = caddf(t, b);
    b = csubf(t, b);
}
__device__ __forceinline__ void fft4(float2& v0, float2& v1, float2& v2, float2& v3) { // natural-order forward DFT-4
    const float2 t0 = caddf(v0, v2), t1 = csubf(v0, v2), t2 = caddf(v1, v3), t3 = mul_mi(csubf(v1, v3));
    v0 = caddf(t0, t2);
    v2 = csubf(t0, t2);
    v1 = caddf(t1, t3);
    v3 = csubf(t1, t3);
}
__device__ __forceinline__ void fft8(float2 (&v)[8]) { // natural-order forward DFT-8 (decimation in time)
    constexpr float h = 0.70710678118654752440f;
    fft4(v[0], v[2], v[4], v[6]);
    fft4(v[1], v[3], v[5], v[7]);
    const float2 o1 = make_float2((v[3].x + v[3].y) * h, (v[3].y - v[3].x) * h);   // * W8^1
    const float2 o2 = mul_mi(v[5]);                                                 // * W8^2
    const float2 o3 = make_float2((v[7].y - v[7].x) * h, (-v[7].x - v[7].y) * h);  // * W8^3
    const float2 e0 = v[0], e1 = v[2], e2 = v[4], e3 = v[6], o0 = v[1];
    v[0] = caddf(e0, o0); v[4] = csubf(e0, o0);
    v[1] = caddf(e1, o1); v[5] = csubf(e1, o1);
    v[2] = caddf(e2, o2); v[6] = csubf(e2, o2);
    v[3] = caddf(e3, o3); v[7] = csubf(e3, o3);
}

struct FftPlanDev {
    int N;
    int npass;
    int radix[16];
    int tf;  // threads per frame
    int fpb; // frames per block
};

struct FftOutputs {
    float* spectrum;  // [frames][N][2]
    float* re;        // complex in: [frames][N]; real in: [frames][N/2] (bins N/2..N-1, fft.hpp:221-227)
    float* im;
    float* mag;       // shifted (complex) / first half (real)
    float* phase;     // final phase when !unwrap
    float* phase_raw; // natural-order raw atan2 (only when unwrap; finished by unwrap_kernel)
    float* mag2;      // natural order
    int    in_db, in_deg, real_input;
};

template <int R>
__device__ __forceinline__ void load_bfly(float2* v, const float2* __restrict__ src, int i, int NB, int p, int step_unit, const float2* __restrict__ tw) {
    const int k = i & (p - 1);
#pragma unroll
    for (int r = 0; r < R; ++r) {
        float2 x = src[i + r * NB];
        if (r > 0 && p > 1) x = cmulf(x, tw[r * k * step_unit]);
        v[r] = x;
    }
}

template <int R>
__device__ __forceinline__ void store_bfly(const float2* v, float2* dst, int i, int p) {
    const int k    = i & (p - 1);
    const int base = (i - k) * R + k;
#pragma unroll
    for (int r = 0; r < R; ++r) dst[base + r * p] = v[r];
}

// One workgroup = fpb frames; tf = max(1, N/8) lanes per frame, 8 points per lane.
__global__ void fft_block_kernel(const float* __restrict__ in, const float* __restrict__ window, const float2* __restrict__ tw, FftPlanDev plan, FftOutputs out,
                                 long n_frames) {
    extern __shared__ __attribute__((aligned(16))) float2 lds[];
    const int  N     = plan.N;
    const int  tf    = plan.tf;
    const int  fl    = threadIdx.x / tf; // frame slot in block
    const int  t     = threadIdx.x - fl * tf;
    const long frame = (long)blockIdx.x * plan.fpb + fl;
    const bool live  = frame < n_frames;
    float2*    buf   = lds + (size_t)fl * N;

    // ---- load + window (fft.hpp:148-162); real input becomes (x*w, 0)
    if (live) {
        if (out.real_input) {
            const float* x = in + frame * N;
            for (int i = t; i < N; i += tf) buf[i] = make_float2(x[i] * (window ? window[i] : 1.f), 0.f);
        } else {
            const float2* x = reinterpret_cast<const float2*>(in) + frame * N;
            for (int i = t; i < N; i += tf) {
                float2 s = x[i];
                if (window) { const float w = window[i]; s.x *= w; s.y *= w; }
                buf[i] = s;
            }
        }
    }
    __syncthreads();

    int p = 1;
    for (int pass = 0; pass < plan.npass; ++pass) {
        const int R  = plan.radix[pass];
        const int NB = N / R;
        const int nb = NB / tf > 0 ? NB / tf : 1;
        const int su = N / (p * R); // twiddle index unit: W_{pR}^{rk} = tw[r*k*su]
        float2    v[8];
        if (R == 8) {
            if (t < NB) load_bfly<8>(v, buf, t, NB, p, su, tw);
        } else if (R == 4) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (b < nb && t + b * tf < NB) load_bfly<4>(v + 4 * b, buf, t + b * tf, NB, p, su, tw);
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (b < nb && t + b * tf < NB) load_bfly<2>(v + 2 * b, buf, t + b * tf, NB, p, su, tw);
        }
        __syncthreads();
        if (R == 8) {
            if (t < NB) { fft8(v); store_bfly<8>(v, buf, t, p); }
        } else if (R == 4) {
#pragma unroll
            for (int b = 0; b < 2; ++b)
                if (b < nb && t + b * tf < NB) { fft4(v[4 * b], v[4 * b + 1], v[4 * b + 2], v[4 * b + 3]); store_bfly<4>(v + 4 * b, buf, t + b * tf, p); }
        } else {
#pragma unroll
            for (int b = 0; b < 4; ++b)
                if (b < nb && t + b * tf < NB) { fft2(v[2 * b], v[2 * b + 1]); store_bfly<2>(v + 2 * b, buf, t + b * tf, p); }
        }
        __syncthreads();
        p *= R;
    }
    if (!live) return;

    // ---- epilogue over natural-order bins (fft.hpp:164-166, fft_common.hpp:20-56, 91-123)
    const int   half  = N / 2;
    const int   nout  = out.real_input ? half : N;
    const float scale = 2.f / (float)N; // hypot * 2 / N
    for (int k = t; k < N; k += tf) {
        const float2 X = buf[k];
        if (out.spectrum) reinterpret_cast<float2*>(out.spectrum)[frame * N + k] = X;
        if (out.mag2) out.mag2[frame * N + k] = fmaf(X.x, X.x, X.y * X.y);
        if (out.real_input) {
            if (k >= half) {
                if (out.re) out.re[frame * half + (k - half)] = X.x;
                if (out.im) out.im[frame * half + (k - half)] = X.y;
            }
        } else {
            if (out.re) out.re[frame * N + k] = X.x;
            if (out.im) out.im[frame * N + k] = X.y;
        }
        if (out.real_input && k >= half) continue; // computeHalfSpectrum: first N/2 bins, never rotated
        const int ko = out.real_input ? k : ((k + half) & (N - 1));
        if (out.mag) {
            float m = hypotf(X.x, X.y) * 2.f / (float)N;
            (void)scale;
            if (out.in_db) m = (m > 0.f) ? 20.f * log10f(m) : -FLT_MAX;
            out.mag[frame * nout + ko] = m;
        }
        if (out.phase || out.phase_raw) {
            float ph = atan2f(X.y, X.x);
            if (out.phase_raw) {
                out.phase_raw[frame * nout + k] = ph;
            } else {
                if (out.in_deg) ph = ph * 180.f * 0.318309886183790671538f;
                out.phase[frame * nout + ko] = ph;
            }
        }
    }
}

// fft_common.hpp:71-89 unwrapPhase + :113-120 (deg, shift).  One workgroup per frame; wrap counts are integers, so a
// parallel prefix sum of the per-bin jump decisions reproduces the sequential loop.
__global__ void unwrap_kernel(const float* __restrict__ raw, float* __restrict__ out, int nout, int in_deg, int shift) {
    extern __shared__ int sc[]; // [blockDim.x]
    const float  pi    = 3.14159265358979323846f;
    const long   frame = blockIdx.x;
    const float* r     = raw + frame * nout;
    float*       o     = out + frame * nout;
    const int    per   = (nout + blockDim.x - 1) / blockDim.x;
    const int    k0    = threadIdx.x * per;
    int          local = 0;
    for (int k = k0; k < k0 + per && k < nout; ++k) {
        if (k == 0) continue;
        const float d = r[k] - r[k - 1];
        local += (d > pi) ? -1 : (d < -pi) ? 1 : 0;
    }
    sc[threadIdx.x] = local;
    __syncthreads();
    for (int off = 1; off < (int)blockDim.x; off <<= 1) { // inclusive Hillis-Steele scan
        const int v = (int)threadIdx.x >= off ? sc[threadIdx.x - off] : 0;
        __syncthreads();
        sc[threadIdx.x] += v;
        __syncthreads();
    }
    int c = sc[threadIdx.x] - local; // exclusive prefix for this lane's first bin
    for (int k = k0; k < k0 + per && k < nout; ++k) {
        if (k > 0) {
            const float d = r[k] - r[k - 1];
            c += (d > pi) ? -1 : (d < -pi) ? 1 : 0;
        }
        float ph = (float)((double)r[k] + (double)c * (double)(2.f * pi));
        if (in_deg) ph = ph * 180.f * 0.318309886183790671538f;
        const int ko = shift ? ((k + nout / 2) % nout) : k;
        o[ko]        = ph;
    }
}

// per-frame {min,max} of the four DataSet signals (fft.hpp:229-232): grid = (frames, 4)
__global__ void ranges_kernel(const float* __restrict__ s0, const float* __restrict__ s1, const float* __restrict__ s2, const float* __restrict__ s3, int nout,
                              float* __restrict__ ranges) {
    __shared__ float smin[256], smax[256];
    const float*     base[4] = {s0, s1, s2, s3};
    const float*     s       = base[blockIdx.y];
    float            mn = FLT_MAX, mx = -FLT_MAX;
    if (s) {
        s += (long)blockIdx.x * nout;
        for (int k = threadIdx.x; k < nout; k += blockDim.x) { const float v = s[k]; mn = fminf(mn, v); mx = fmaxf(mx, v); }
    }
    smin[threadIdx.x] = mn;
    smax[threadIdx.x] = mx;
    __syncthreads();
    for (int off = blockDim.x / 2; off > 0; off >>= 1) {
        if ((int)threadIdx.x < off) {
            smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + off]);
            smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + off]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        ranges[((long)blockIdx.x * 4 + blockIdx.y) * 2 + 0] = s ? smin[0] : 0.f;
        ranges[((long)blockIdx.x * 4 + blockIdx.y) * 2 + 1] = s ? smax[0] : 0.f;
    }
}

} // namespace gr4

using namespace gr4;

struct gr4hip_fft {
    int          in_dtype = GR4HIP_C32;
    size_t       N        = 0;
    int          window   = GR4HIP_WIN_HANN;
    int          flags    = 0;
    FftPlanDev   plan{};
    DeviceBuffer d_window, d_tw, d_phase_raw;
};

namespace gr4 {
// shared with chain.hip
int fft_build_plan(size_t N, FftPlanDev* plan) {
    if (!is_pow2(N) || N < 2) { set_error("fft: size %zu is not a power of two >= 2 (device path)", N); return GR4HIP_UNSUPPORTED; }
    if (N > 8192) { set_error("fft: size %zu exceeds the single-workgroup LDS path (max 8192)", N); return GR4HIP_UNSUPPORTED; }
    plan->N  = (int)N;
    int l    = ilog2(N);
    int np   = 0;
    while (l >= 3) { plan->radix[np++] = 8; l -= 3; }
    if (l == 2) plan->radix[np++] = 4;
    if (l == 1) plan->radix[np++] = 2;
    plan->npass = np;
    plan->tf    = N >= 8 ? (int)(N / 8) : 1;
    plan->fpb   = plan->tf >= 64 ? 1 : 64 / plan->tf;
    return GR4HIP_OK;
}

int fft_upload_twiddles(size_t N, DeviceBuffer* buf) {
    std::vector<float> tw(2 * N);
    for (size_t k = 0; k < N; ++k) {
        const double a = -2.0 * M_PI * (double)k / (double)N;
        tw[2 * k]     = (float)std::cos(a);
        tw[2 * k + 1] = (float)std::sin(a);
    }
    int rc = buf->ensure(tw.size() * sizeof(float));
    if (rc) return rc;
    GR4_HIP_TRY(hipMemcpy(buf->ptr, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
    return GR4HIP_OK;
}

int fft_launch(const FftPlanDev& plan, const float* d_in, const float* d_window, const float2* d_tw, const FftOutputs& o, long n_frames, hipStream_t st) {
    const size_t lds  = (size_t)plan.fpb * plan.N * sizeof(float2);
    const int    bs   = plan.tf * plan.fpb;
    const long   grid = ceil_div(n_frames, (long)plan.fpb);
    if (lds > 48 * 1024) GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(fft_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(fft_block_kernel, dim3((unsigned)grid), dim3(bs), lds, st, d_in, d_window, d_tw, plan, o, n_frames);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}
} // namespace gr4

extern "C" {

int gr4hip_fft_create(gr4hip_fft_t** out, int in_dtype, size_t fft_size, int window, int flags) {
    GR4_REQUIRE(out, "fft: null output handle");
    GR4_REQUIRE(in_dtype == GR4HIP_F32 || in_dtype == GR4HIP_C32, "fft: input dtype must be F32 or C32 (got %d)", in_dtype);
    GR4_REQUIRE(window >= GR4HIP_WIN_NONE && window <= GR4HIP_WIN_KAISER, "fft: unknown window %d", window);
    auto* f = new (std::nothrow) gr4hip_fft();
    GR4_REQUIRE(f, "out of host memory");
    f->in_dtype = in_dtype;
    f->N        = fft_size;
    f->window   = window;
    f->flags    = flags;
    int rc      = fft_build_plan(fft_size, &f->plan);
    if (!rc) rc = fft_upload_twiddles(fft_size, &f->d_tw);
    if (!rc && window != GR4HIP_WIN_NONE && window != GR4HIP_WIN_RECTANGULAR) {
        std::vector<float> w(fft_size);
        rc = make_window(window, w.data(), fft_size, 1.6f); // fft.hpp:141: create(_window, _windowType) -> default beta
        if (!rc) rc = f->d_window.ensure(fft_size * sizeof(float));
        if (!rc) { hipError_t e = hipMemcpy(f->d_window.ptr, w.data(), fft_size * sizeof(float), hipMemcpyHostToDevice); if (e != hipSuccess) { set_error("window upload: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
    }
    if (rc) { delete f; return rc; }
    *out = f;
    return GR4HIP_OK;
}

static int fft_run(gr4hip_fft_t* f, const void* d_in, size_t n_frames, FftOutputs o, float* d_phase_final, float* d_ranges, gr4hip_stream_t stream) {
    GR4_REQUIRE(f, "fft: null handle");
    if (n_frames == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in, "fft: null input");
    hipStream_t st   = as_stream(stream);
    o.real_input     = f->in_dtype == GR4HIP_F32;
    o.in_db          = (f->flags & GR4HIP_FFT_OUTPUT_IN_DB) != 0;
    o.in_deg         = (f->flags & GR4HIP_FFT_OUTPUT_IN_DEG) != 0;
    const int  nout  = o.real_input ? (int)f->N / 2 : (int)f->N;
    const bool unwrap = (f->flags & GR4HIP_FFT_UNWRAP_PHASE) && d_phase_final;
    if (unwrap) {
        int rc = f->d_phase_raw.ensure(n_frames * nout * sizeof(float));
        if (rc) return rc;
        o.phase_raw = static_cast<float*>(f->d_phase_raw.ptr);
        o.phase     = nullptr;
    } else {
        o.phase     = d_phase_final;
        o.phase_raw = nullptr;
    }
    int rc = fft_launch(f->plan, static_cast<const float*>(d_in), static_cast<const float*>(f->d_window.ptr), static_cast<const float2*>(f->d_tw.ptr), o,
                        (long)n_frames, st);
    if (rc) return rc;
    if (unwrap) {
        const int bs = nout >= 256 ? 256 : 64;
        hipLaunchKernelGGL(unwrap_kernel, dim3((unsigned)n_frames), dim3(bs), bs * sizeof(int), st, (const float*)o.phase_raw, d_phase_final, nout, o.in_deg,
                           o.real_input ? 0 : 1);
        GR4_LAUNCH_CHECK();
    }
    if (d_ranges) {
        hipLaunchKernelGGL(ranges_kernel, dim3((unsigned)n_frames, 4), dim3(256), 0, st, (const float*)o.mag, (const float*)d_phase_final, (const float*)o.re,
                           (const float*)o.im, nout, d_ranges);
        GR4_LAUNCH_CHECK();
    }
    return GR4HIP_OK;
}

int gr4hip_fft_process(gr4hip_fft_t* f, const void* d_in, size_t n_frames, float* d_mag, float* d_phase, float* d_re, float* d_im, float* d_ranges,
                       gr4hip_stream_t stream) {
    FftOutputs o{};
    o.mag = d_mag;
    o.re  = d_re;
    o.im  = d_im;
    return fft_run(f, d_in, n_frames, o, d_phase, d_ranges, stream);
}

int gr4hip_fft_spectrum(gr4hip_fft_t* f, const void* d_in, size_t n_frames, float* d_spectrum, gr4hip_stream_t stream) {
    GR4_REQUIRE(d_spectrum || n_frames == 0, "fft_spectrum: null output");
    FftOutputs o{};
    o.spectrum = d_spectrum;
    return fft_run(f, d_in, n_frames, o, nullptr, nullptr, stream);
}

int gr4hip_fft_mag2(gr4hip_fft_t* f, const void* d_in, size_t n_frames, float* d_mag2, gr4hip_stream_t stream) {
    GR4_REQUIRE(d_mag2 || n_frames == 0, "fft_mag2: null output");
    FftOutputs o{};
    o.mag2 = d_mag2;
    return fft_run(f, d_in, n_frames, o, nullptr, nullptr, stream);
}

int gr4hip_fft_destroy(gr4hip_fft_t* f) { delete f; return GR4HIP_OK; }

} // extern "C"
