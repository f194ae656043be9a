// iir.hip -- IIR section cascades on gfx950: exact parallel-in-time evaluation of a sequential recurrence.
//
// Replaces gr::filter::Filter<float>::processOne (std::accumulate over sections of detail::computeFilter,
// algorithm/.../filter/FilterTool.hpp:116-158, 244-246) and gr::filter::iir_filter<float,form>::processOne
// (blocks/filter/.../time_domain_filter.hpp:89-121).  The cascade is one LTI system with state s (M floats, the
// direct-form-II delay lines of all sections).  For a chunk of L samples:  s_end = Phi_L * s_start + z  with z the
// zero-state response end state, so chunk start states follow from a (matrix) prefix scan:
//   pass Z : every lane runs its L-sample chunk from zero state (tile staged in LDS, conflict-free rows) -> z_c;
//            in-block Kogge-Stone scan with host-precomputed Phi_{L*2^k} gives the block's zero-state end state;
//   pass B : one wave chains the block states  T_{b+1} = Phi_B * T_b + Z_b  (the only sequential step: n/8192 steps);
//   pass Y : same in-block scan seeded with T_b gives every chunk's true start state; the chunk is re-run from it,
//            results go back to the LDS tile and leave with coalesced stores.
// Bound: FP32 FMA latency chains, 2 recurrence passes per sample (DESIGN.md "iir_cascade").
#include "common.hpp"

#include <cmath>

namespace gr4 {

constexpr int kIirL      = 32;  // samples per lane chunk
constexpr int kIirBS     = 256; // lanes per block
constexpr int kIirMaxSec = 8;
constexpr int kIirMaxM   = 16;  // total state floats
constexpr int kIirRounds = 8;   // log2(kIirBS)

template <int ORD>
struct IirCoef {
    int   nsec;
    float b[kIirMaxSec][ORD + 1];
    float a[kIirMaxSec][ORD + 1]; // a[.][0] unused (== 1)
};

// one sample through the cascade, direct form II per section (FilterTool.hpp:130-141)
template <int ORD>
__device__ __forceinline__ float iir_step(const IirCoef<ORD>& c, float (&st)[kIirMaxSec][ORD], float x) {
#pragma unroll
    for (int s = 0; s < kIirMaxSec; ++s) {
        if (s < c.nsec) {
            float w = x;
#pragma unroll
            for (int j = 0; j < ORD; ++j) w = fmaf(-c.a[s][j + 1], st[s][j], w);
            float y = c.b[s][0] * w;
#pragma unroll
            for (int j = 0; j < ORD; ++j) y = fmaf(c.b[s][j + 1], st[s][j], y);
#pragma unroll
            for (int j = ORD - 1; j > 0; --j) st[s][j] = st[s][j - 1];
            st[s][0] = w;
            x        = y;
        }
    }
    return x;
}

// inclusive scan over the block's chunks:  sv[c] <- sum_{i<=c} Phi_L^{c-i} sv[i]   (sv holds M floats per lane)
__device__ __forceinline__ void iir_block_scan(float* sv, const float* __restrict__ phi /*[rounds][M][M]*/, int M) {
    const int c = threadIdx.x;
    for (int k = 0; k < kIirRounds; ++k) {
        const int    off = 1 << k;
        float        tmp[kIirMaxM];
        const float* P = phi + (size_t)k * M * M;
        if (c >= off) {
#pragma unroll
            for (int i = 0; i < kIirMaxM; ++i) {
                if (i < M) {
                    float acc = 0.f;
                    for (int j = 0; j < M; ++j) acc = fmaf(P[i * M + j], sv[(c - off) * kIirMaxM + j], acc);
                    tmp[i] = acc;
                }
            }
        }
        __syncthreads();
        if (c >= off) {
#pragma unroll
            for (int i = 0; i < kIirMaxM; ++i)
                if (i < M) sv[c * kIirMaxM + i] += tmp[i];
        }
        __syncthreads();
    }
}

template <int ORD>
__device__ __forceinline__ void iir_stage_tile(float* tile, const float* __restrict__ x, long base, long n) {
    for (int s = threadIdx.x; s < kIirBS * kIirL; s += kIirBS) {
        const long i = base + s;
        tile[(s / kIirL) * (kIirL + 1) + (s % kIirL)] = i < n ? x[i] : 0.f;
    }
}

// pass Z: z_c per chunk (global, [chunks][M]) and the block's zero-state end state Zb[block][M]
template <int ORD>
__global__ __launch_bounds__(kIirBS) void iir_pass_z(const float* __restrict__ x, long n, IirCoef<ORD> coef, const float* __restrict__ phi, float* __restrict__ zc,
                                                      float* __restrict__ zb) {
    __shared__ float tile[kIirBS * (kIirL + 1)];
    __shared__ float sv[kIirBS * kIirMaxM];
    const int  M    = coef.nsec * ORD;
    const long base = (long)blockIdx.x * kIirBS * kIirL;
    iir_stage_tile<ORD>(tile, x, base, n);
    __syncthreads();
    float st[kIirMaxSec][ORD];
#pragma unroll
    for (int s = 0; s < kIirMaxSec; ++s)
#pragma unroll
        for (int j = 0; j < ORD; ++j) st[s][j] = 0.f;
    const float* row = tile + threadIdx.x * (kIirL + 1);
#pragma unroll 4
    for (int i = 0; i < kIirL; ++i) (void)iir_step<ORD>(coef, st, row[i]);
    const long chunk = (long)blockIdx.x * kIirBS + threadIdx.x;
#pragma unroll
    for (int s = 0; s < kIirMaxSec; ++s)
#pragma unroll
        for (int j = 0; j < ORD; ++j)
            if (s < coef.nsec) {
                sv[threadIdx.x * kIirMaxM + s * ORD + j] = st[s][j];
                zc[chunk * kIirMaxM + s * ORD + j]        = st[s][j];
            }
    __syncthreads();
    iir_block_scan(sv, phi, M);
    if (threadIdx.x < M) zb[(long)blockIdx.x * kIirMaxM + threadIdx.x] = sv[(kIirBS - 1) * kIirMaxM + threadIdx.x];
}

// pass B: T_0 = carried state; T_{b+1} = Phi_B T_b + Zb[b].  Writes T_b (state at the START of block b).
__global__ void iir_pass_b(const float* __restrict__ state0, const float* __restrict__ phiB, const float* __restrict__ zb, float* __restrict__ tb, long nblocks, int M) {
    __shared__ float T[kIirMaxM];
    const int        i = threadIdx.x;
    if (i < M) T[i] = state0[i];
    __syncthreads();
    for (long b = 0; b < nblocks; ++b) {
        float nv = 0.f;
        if (i < M) {
            tb[b * kIirMaxM + i] = T[i];
            nv = zb[b * kIirMaxM + i];
            for (int j = 0; j < M; ++j) nv = fmaf(phiB[i * M + j], T[j], nv);
        }
        __syncthreads();
        if (i < M) T[i] = nv;
        __syncthreads();
    }
}

// pass Y: true start state per chunk, re-run, coalesced store.  The lane owning the last sample stores the carried state.
template <int ORD>
__global__ __launch_bounds__(kIirBS) void iir_pass_y(const float* __restrict__ x, float* __restrict__ y, long n, IirCoef<ORD> coef, const float* __restrict__ phi,
                                                      const float* __restrict__ zc, const float* __restrict__ tb, float* __restrict__ state_out) {
    __shared__ float tile[kIirBS * (kIirL + 1)];
    __shared__ float sv[kIirBS * kIirMaxM];
    const int  M     = coef.nsec * ORD;
    const int  c     = threadIdx.x;
    const long base  = (long)blockIdx.x * kIirBS * kIirL;
    const long chunk = (long)blockIdx.x * kIirBS + c;
    iir_stage_tile<ORD>(tile, x, base, n);
    // seed: I_0 = Phi_L * T_b + z_0, I_c = z_c
    const float* Tb = tb + (long)blockIdx.x * kIirMaxM;
#pragma unroll
    for (int i = 0; i < kIirMaxM; ++i) {
        if (i < M) {
            float v = zc[chunk * kIirMaxM + i];
            if (c == 0)
                for (int j = 0; j < M; ++j) v = fmaf(phi[i * M + j], Tb[j], v); // round-0 matrix == Phi_L
            sv[c * kIirMaxM + i] = v;
        }
    }
    __syncthreads();
    iir_block_scan(sv, phi, M);
    float st[kIirMaxSec][ORD];
#pragma unroll
    for (int s = 0; s < kIirMaxSec; ++s)
#pragma unroll
        for (int j = 0; j < ORD; ++j) st[s][j] = (s < coef.nsec) ? (c == 0 ? Tb[s * ORD + j] : sv[(c - 1) * kIirMaxM + s * ORD + j]) : 0.f;
    float*     row  = tile + c * (kIirL + 1);
    const long cbeg = base + (long)c * kIirL;
    const int  len  = (int)(n - cbeg < kIirL ? (n - cbeg < 0 ? 0 : n - cbeg) : kIirL);
    for (int i = 0; i < len; ++i) row[i] = iir_step<ORD>(coef, st, row[i]);
    if (len > 0 && cbeg + len == n) { // this lane consumed the last sample of the span
#pragma unroll
        for (int s = 0; s < kIirMaxSec; ++s)
#pragma unroll
            for (int j = 0; j < ORD; ++j)
                if (s < coef.nsec) state_out[s * ORD + j] = st[s][j];
    }
    __syncthreads();
    for (int s = threadIdx.x; s < kIirBS * kIirL; s += kIirBS) {
        const long i = base + s;
        if (i < n) y[i] = tile[(s / kIirL) * (kIirL + 1) + (s % kIirL)];
    }
}

} // namespace gr4

using namespace gr4;

struct gr4hip_iir {
    int                 form = GR4HIP_DF_II;
    int                 nsec = 0, ord = 2, M = 0;
    std::vector<double> b, a; // [nsec][ord+1]
    DeviceBuffer        d_phi;   // [rounds][M][M] then Phi_B [M][M]
    DeviceBuffer        d_state[2];
    int                 cur = 0;
    DeviceBuffer        d_zc, d_zb, d_tb;
};

// host double-precision cascade step (same recurrence) used to build the propagation matrices
static void host_step(const gr4hip_iir* f, std::vector<double>& st, double x) {
    const int O = f->ord;
    for (int s = 0; s < f->nsec; ++s) {
        double w = x;
        for (int j = 0; j < O; ++j) w -= f->a[s * (O + 1) + j + 1] * st[s * O + j];
        double y = f->b[s * (O + 1)] * w;
        for (int j = 0; j < O; ++j) y += f->b[s * (O + 1) + j + 1] * st[s * O + j];
        for (int j = O - 1; j > 0; --j) st[s * O + j] = st[s * O + j - 1];
        st[s * O] = w;
        x         = y;
    }
}

static void host_phi(const gr4hip_iir* f, long steps, float* out /*[M][M]*/) {
    const int M = f->M;
    for (int j = 0; j < M; ++j) {
        std::vector<double> st(M, 0.0);
        st[j] = 1.0;
        for (long t = 0; t < steps; ++t) host_step(f, st, 0.0);
        for (int i = 0; i < M; ++i) out[i * M + j] = (float)st[i];
    }
}

template <int ORD>
static IirCoef<ORD> make_coef(const gr4hip_iir* f) {
    IirCoef<ORD> c{};
    c.nsec = f->nsec;
    for (int s = 0; s < f->nsec; ++s)
        for (int j = 0; j <= ORD; ++j) {
            c.b[s][j] = (float)f->b[s * (ORD + 1) + j];
            c.a[s][j] = (float)f->a[s * (ORD + 1) + j];
        }
    return c;
}

template <int ORD>
static int iir_run(gr4hip_iir* f, const float* x, float* y, long n, hipStream_t st) {
    const long nblocks = ceil_div(n, (long)kIirBS * kIirL);
    int        rc      = f->d_zc.ensure((size_t)nblocks * kIirBS * kIirMaxM * sizeof(float));
    if (!rc) rc = f->d_zb.ensure((size_t)nblocks * kIirMaxM * sizeof(float));
    if (!rc) rc = f->d_tb.ensure((size_t)nblocks * kIirMaxM * sizeof(float));
    if (rc) return rc;
    const IirCoef<ORD> coef = make_coef<ORD>(f);
    const float*       phi  = static_cast<const float*>(f->d_phi.ptr);
    const float*       phiB = phi + (size_t)kIirRounds * f->M * f->M;
    hipLaunchKernelGGL(iir_pass_z<ORD>, dim3((unsigned)nblocks), dim3(kIirBS), 0, st, x, n, coef, phi, (float*)f->d_zc.ptr, (float*)f->d_zb.ptr);
    GR4_LAUNCH_CHECK();
    hipLaunchKernelGGL(iir_pass_b, dim3(1), dim3(64), 0, st, (const float*)f->d_state[f->cur].ptr, phiB, (const float*)f->d_zb.ptr, (float*)f->d_tb.ptr, nblocks, f->M);
    GR4_LAUNCH_CHECK();
    hipLaunchKernelGGL(iir_pass_y<ORD>, dim3((unsigned)nblocks), dim3(kIirBS), 0, st, x, y, n, coef, phi, (const float*)f->d_zc.ptr, (const float*)f->d_tb.ptr,
                       (float*)f->d_state[f->cur ^ 1].ptr);
    GR4_LAUNCH_CHECK();
    f->cur ^= 1;
    return GR4HIP_OK;
}

extern "C" {

int gr4hip_iir_create(gr4hip_iir_t** out, int form, size_t nsections, const float* h_b, size_t nb, const float* h_a, size_t na) {
    GR4_REQUIRE(out, "iir: null output handle");
    GR4_REQUIRE(form >= GR4HIP_DF_I && form <= GR4HIP_DF_II_TRANSPOSED, "iir: unknown form %d", form);
    GR4_REQUIRE(nsections >= 1 && h_b && h_a && nb >= 1 && na >= 1, "iir: need >= 1 section and non-empty b, a");
    const size_t order = std::max(nb, na) - 1;
    const int    ord   = order <= 2 ? 2 : 4;
    if (order > 4 || nsections > (size_t)kIirMaxSec || nsections * ord > (size_t)kIirMaxM) {
        set_error("iir: %zu sections of order %zu exceed the device path (order <= 4, sections*order <= %d)", nsections, order, kIirMaxM);
        return GR4HIP_UNSUPPORTED;
    }
    auto* f = new (std::nothrow) gr4hip_iir();
    GR4_REQUIRE(f, "out of host memory");
    f->form = form;
    f->nsec = (int)nsections;
    f->ord  = ord;
    f->M    = f->nsec * ord;
    f->b.assign((size_t)f->nsec * (ord + 1), 0.0);
    f->a.assign((size_t)f->nsec * (ord + 1), 0.0);
    for (int s = 0; s < f->nsec; ++s) {
        for (size_t j = 0; j < nb; ++j) f->b[s * (ord + 1) + j] = h_b[s * nb + j];
        for (size_t j = 0; j < na; ++j) f->a[s * (ord + 1) + j] = h_a[s * na + j];
        f->a[s * (ord + 1)] = 1.0; // a[0] is assumed 1 (time_domain_filter.hpp:95,102)
    }
    const size_t       mm = (size_t)f->M * f->M;
    std::vector<float> phi((kIirRounds + 1) * mm);
    for (int k = 0; k < kIirRounds; ++k) host_phi(f, (long)kIirL << k, phi.data() + k * mm);
    host_phi(f, (long)kIirL * kIirBS, phi.data() + kIirRounds * mm);
    int rc = f->d_phi.ensure(phi.size() * sizeof(float));
    if (!rc) { hipError_t e = hipMemcpy(f->d_phi.ptr, phi.data(), phi.size() * sizeof(float), hipMemcpyHostToDevice); if (e != hipSuccess) { set_error("iir: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
    for (int k = 0; k < 2 && !rc; ++k) rc = f->d_state[k].ensure(kIirMaxM * sizeof(float));
    if (rc) { delete f; return rc; }
    rc = gr4hip_iir_reset(f);
    if (rc) { delete f; return rc; }
    *out = f;
    return GR4HIP_OK;
}

int gr4hip_iir_reset(gr4hip_iir_t* f) {
    GR4_REQUIRE(f, "iir_reset: null handle");
    for (int k = 0; k < 2; ++k) GR4_HIP_TRY(hipMemset(f->d_state[k].ptr, 0, kIirMaxM * sizeof(float)));
    f->cur = 0;
    return GR4HIP_OK;
}

int gr4hip_iir_process(gr4hip_iir_t* f, const float* d_in, size_t n, float* d_out, gr4hip_stream_t stream) {
    GR4_REQUIRE(f, "iir_process: null handle");
    if (n == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in && d_out, "iir_process: null device pointer");
    return f->ord == 2 ? iir_run<2>(f, d_in, d_out, (long)n, as_stream(stream)) : iir_run<4>(f, d_in, d_out, (long)n, as_stream(stream));
}

int gr4hip_iir_destroy(gr4hip_iir_t* f) { delete f; return GR4HIP_OK; }

} // extern "C"
