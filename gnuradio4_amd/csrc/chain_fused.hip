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
// One workgroup (256 lanes) = one frame.  Stockham radix 32 x 16 x 16; every lane holds 32 complex points.
// LDS exchange layouts are padded (rows of 272 / 513 float2) so all ds_read_b64 / ds_write_b64 are conflict-free.
#include "common.hpp"

#include <cmath>
#include <complex>

namespace gr4 {

constexpr int kN     = 8192;
constexpr int kT     = 256;  // lanes per workgroup
constexpr int kTail  = 255;  // ntaps - 1 (taps are zero-padded to 256)
constexpr int kRowA  = 272;  // pass-A -> pass-B exchange: S[r'][t], row pitch 272 float2 (544 dwords = 32 mod 64 banks)
constexpr int kRowB  = 513;  // pass-B -> pass-C exchange: S[c][i3], row pitch 513 float2 (= 1 mod 16 -> 16 lanes spread over 32 banks)
constexpr int kSLen  = 32 * kRowA; // 8704 float2 >= 16 * 513

__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)); }
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ float2 mulmi(float2 a) { return make_float2(a.y, -a.x); }

__device__ __forceinline__ void bfly4(float2& v0, float2& v1, float2& v2, float2& v3) {
    const float2 t0 = cadd(v0, v2), t1 = csub(v0, v2), t2 = cadd(v1, v3), t3 = mulmi(csub(v1, v3));
    v0 = cadd(t0, t2); v2 = csub(t0, t2); v1 = cadd(t1, t3); v3 = csub(t1, t3);
}

// multiply by W_16^m with compile-time m
template <int M>
__device__ __forceinline__ float2 mul_w16(float2 a) {
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, h = 0.70710678118654752440f;
    if constexpr (M == 0) return a;
    else if constexpr (M == 1) return make_float2(fmaf(a.y, s1, a.x * c1), fmaf(-a.x, s1, a.y * c1));  // (c1 - i s1)
    else if constexpr (M == 2) return make_float2((a.x + a.y) * h, (a.y - a.x) * h);
    else if constexpr (M == 3) return make_float2(fmaf(a.y, c1, a.x * s1), fmaf(-a.x, c1, a.y * s1));  // (s1 - i c1)
    else if constexpr (M == 4) return mulmi(a);
    else { static_assert(M == 6, "unsupported W_16 power"); return make_float2((a.y - a.x) * h, (-a.x - a.y) * h); }
}

// in-place 16-point forward DFT on v[0], v[ST], ..., v[15 ST].  Result order: X[m] sits at slot perm16(m).
__host__ __device__ constexpr int perm16(int m) { return 4 * (m & 3) + (m >> 2); }

template <int ST>
__device__ __forceinline__ void fft16(float2* v) {
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) bfly4(v[(n2)*ST], v[(n2 + 4) * ST], v[(n2 + 8) * ST], v[(n2 + 12) * ST]);
    // a[n2][k1] is at slot n2 + 4 k1; twiddle W_16^{n2 k1}
    v[5 * ST]  = mul_w16<1>(v[5 * ST]);   // n2=1,k1=1
    v[9 * ST]  = mul_w16<2>(v[9 * ST]);   // n2=1,k1=2
    v[13 * ST] = mul_w16<3>(v[13 * ST]);  // n2=1,k1=3
    v[6 * ST]  = mul_w16<2>(v[6 * ST]);   // n2=2,k1=1
    v[10 * ST] = mul_w16<4>(v[10 * ST]);  // n2=2,k1=2
    v[14 * ST] = mul_w16<6>(v[14 * ST]);  // n2=2,k1=3
    v[7 * ST]  = mul_w16<3>(v[7 * ST]);   // n2=3,k1=1
    v[11 * ST] = mul_w16<6>(v[11 * ST]);  // n2=3,k1=2
    {   // n2=3,k1=3: W_16^9 = (-c1, +s1)
        constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
        const float2 a = v[15 * ST];
        v[15 * ST] = make_float2(fmaf(-a.y, s1, -a.x * c1), fmaf(a.x, s1, -a.y * c1));
    }
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) bfly4(v[(4 * k1) * ST], v[(4 * k1 + 1) * ST], v[(4 * k1 + 2) * ST], v[(4 * k1 + 3) * ST]);
}

// W_32^k = (cos(2 pi k / 32), -sin(2 pi k / 32)); k is a compile-time constant after unrolling, so the switch folds
__device__ __forceinline__ float2 w32(int k) {
    switch (k) {
    case 0: return make_float2(1.0f, -0.0f);
    case 1: return make_float2(0.98078528040323044913f, -0.19509032201612826785f);
    case 2: return make_float2(0.92387953251128675613f, -0.38268343236508977173f);
    case 3: return make_float2(0.83146961230254523708f, -0.55557023301960222474f);
    case 4: return make_float2(0.70710678118654752440f, -0.70710678118654752440f);
    case 5: return make_float2(0.55557023301960222474f, -0.83146961230254523708f);
    case 6: return make_float2(0.38268343236508977173f, -0.92387953251128675613f);
    case 7: return make_float2(0.19509032201612826785f, -0.98078528040323044913f);
    case 8: return make_float2(0.0f, -1.0f);
    case 9: return make_float2(-0.19509032201612826785f, -0.98078528040323044913f);
    case 10: return make_float2(-0.38268343236508977173f, -0.92387953251128675613f);
    case 11: return make_float2(-0.55557023301960222474f, -0.83146961230254523708f);
    case 12: return make_float2(-0.70710678118654752440f, -0.70710678118654752440f);
    case 13: return make_float2(-0.83146961230254523708f, -0.55557023301960222474f);
    case 14: return make_float2(-0.92387953251128675613f, -0.38268343236508977173f);
    default: return make_float2(-0.98078528040323044913f, -0.19509032201612826785f);
    }
}

struct ChainFdArgs {
    const float2* x;      // frames * 8192 samples
    const float2* hist;   // 256 samples preceding x (hist[h] = x[-256 + h])
    const float2* H;      // FFT_8192 of the zero-padded taps
    const float2* twB;    // [16][32]  W_512^{r k}
    const float2* twC;    // [16][512] W_8192^{r i3}
    const float*  taps;   // 256 (zero padded)
    float*        out;    // frames * 8192 mag2
    long          n_frames;
};

// pass B (p = 32, radix 16): butterfly (c, k) gathers S1[k][c + 16 r]
__device__ __forceinline__ void passB_load(const float2* S, float2 (&w)[16], int c, int k) {
#pragma unroll
    for (int r = 0; r < 16; ++r) w[r] = S[k * kRowA + c + 16 * r];
}
__device__ __forceinline__ void passB_compute_store(float2* S, float2 (&w)[16], const float2* __restrict__ twB, int c, int k) {
#pragma unroll
    for (int r = 1; r < 16; ++r) w[r] = cmul(w[r], twB[r * 32 + k]);
    fft16<1>(w);
#pragma unroll
    for (int q = 0; q < 16; ++q) S[c * kRowB + 32 * q + k] = w[perm16(q)];
}
// pass C (p = 512, radix 16): butterfly i3 gathers S2[r][i3]; result X[i3 + 512 q] at slot perm16(q)
__device__ __forceinline__ void passC(const float2* S, float2 (&g)[16], const float2* __restrict__ twC, int i3) {
#pragma unroll
    for (int r = 0; r < 16; ++r) g[r] = S[r * kRowB + i3];
#pragma unroll
    for (int r = 1; r < 16; ++r) g[r] = cmul(g[r], twC[r * 512 + i3]);
    fft16<1>(g);
}
// FFT(y_f)[k] = H[k] X[k] + E[k];  mag2 = |.|^2   (k = i3 + 512 q)
__device__ __forceinline__ void combine_store(float* __restrict__ out, const float2* __restrict__ H, const float2 (&X)[16], const float2 (&E)[16], int i3) {
#pragma unroll
    for (int q = 0; q < 16; ++q) {
        const int    k = i3 + 512 * q;
        const float2 Y = cadd(cmul(H[k], X[perm16(q)]), E[perm16(q)]);
        out[k]         = fmaf(Y.x, Y.x, Y.y * Y.y);
    }
}

__global__ __launch_bounds__(kT, 2) void chain_fd_kernel(ChainFdArgs a) {
    extern __shared__ __attribute__((aligned(16))) float2 smem[];
    float2* S  = smem;              // kSLen
    float2* dl = smem + kSLen;      // 256: d[i]
    float2* el = dl + 256;          // 256: e[n]
    float*  hl = reinterpret_cast<float*>(el + 256); // 256 taps

    const int     t = threadIdx.x;
    const long    f = blockIdx.x;
    const float2* x = a.x + f * kN;

    // ------------------------------------------------------------------ pass A: 32-point DFT down the 256 columns
    float2 v[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) v[r] = x[t + 256 * r];
    hl[t] = a.taps[t];
    {   // d[t-1] = (previous frame's sample at the same tail position) - (this frame's): only lanes 1..255
        const float2 prev = (f > 0) ? x[t + 256 * 31 - kN] : a.hist[t];
        dl[(t + 255) & 255] = (t > 0) ? csub(prev, v[31]) : make_float2(0.f, 0.f); // lane 0 clears slot 255
    }
    // X[k1 + 16 k2] = E16[k1] + (-1)^k2 W_32^k1 O16[k1]
    fft16<2>(v);
    fft16<2>(v + 1);
#pragma unroll
    for (int k1 = 0; k1 < 16; ++k1) {
        const float2 e  = v[2 * perm16(k1)];
        const float2 o  = v[2 * perm16(k1) + 1];
        const float2 wo = k1 == 0 ? o : cmul(o, w32(k1));
        S[k1 * kRowA + t]        = cadd(e, wo);
        S[(k1 + 16) * kRowA + t] = csub(e, wo);
    }
    __syncthreads();

    // ------------------------------------------------------------------ e[n] = sum_{i=n}^{254} d[i] * b[255 + n - i]
    {
        float2 acc = make_float2(0.f, 0.f);
        for (int i = 0; i < kTail; ++i) {
            const float2 dv = dl[i];                          // uniform address: LDS broadcast
            const int    hi = 255 + t - i;
            const float  hv = (i >= t && hi < 256) ? hl[hi & 255] : 0.f;
            acc.x = fmaf(dv.x, hv, acc.x);
            acc.y = fmaf(dv.y, hv, acc.y);
        }
        el[t] = (t < kTail) ? acc : make_float2(0.f, 0.f);
    }

    // ------------------------------------------------------------------ X pass B (p = 32, radix 16) and pass C (p = 512, radix 16)
    // butterfly u = t + 256 h  ->  (c, k) = (u & 15, u >> 4) for pass B;  i3 = u for pass C
    const int c0 = t & 15, k0 = t >> 4, k1 = k0 + 16;
    float2    X0[16], X1[16], w0[16], w1[16];
    passB_load(S, w0, c0, k0);
    passB_load(S, w1, c0, k1);
    __syncthreads();
    passB_compute_store(S, w0, a.twB, c0, k0);
    passB_compute_store(S, w1, a.twB, c0, k1);
    __syncthreads();
    passC(S, X0, a.twC, t);
    passC(S, X1, a.twC, t + 256);
    __syncthreads(); // every lane has consumed S; e[] is complete

    // ------------------------------------------------------------------ E: pass A is a broadcast, then the same passes B and C
#pragma unroll
    for (int r = 0; r < 16; ++r) w0[r] = el[c0 + 16 * r];
#pragma unroll
    for (int r = 0; r < 16; ++r) w1[r] = w0[r];
    passB_compute_store(S, w0, a.twB, c0, k0);
    passB_compute_store(S, w1, a.twB, c0, k1);
    __syncthreads();
    float* out = a.out + f * kN;
    passC(S, w0, a.twC, t);
    combine_store(out, a.H, X0, w0, t);
    passC(S, w1, a.twC, t + 256);
    combine_store(out, a.H, X1, w1, t + 256);
}

int chain_fused_reset(struct ChainFused* c);
struct ChainFused {
    size_t       ntaps = 0;
    DeviceBuffer d_H, d_twB, d_twC, d_taps, d_hist;
};

int chain_fused_supported(size_t ntaps, size_t fft_size, int window, int algo) {
    if (algo != GR4HIP_CHAIN_FUSED_FD) return 0;
    return fft_size == (size_t)kN && ntaps >= 1 && ntaps <= 256 && (window == GR4HIP_WIN_NONE || window == GR4HIP_WIN_RECTANGULAR);
}

template <typename T>
static int upload(DeviceBuffer& b, const std::vector<T>& h) {
    int rc = b.ensure(h.size() * sizeof(T));
    if (rc) return rc;
    GR4_HIP_TRY(hipMemcpy(b.ptr, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
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
    if (!rc) rc = c->d_hist.ensure(256 * sizeof(float2));
    if (!rc) rc = chain_fused_reset(c);
    if (rc) { delete c; return rc; }
    *out = c;
    return GR4HIP_OK;
}

int chain_fused_reset(ChainFused* c) {
    GR4_HIP_TRY(hipMemset(c->d_hist.ptr, 0, 256 * sizeof(float2)));
    return GR4HIP_OK;
}

int chain_fused_process(ChainFused* c, const float* d_in, size_t n_frames, float* d_mag2, hipStream_t st) {
    ChainFdArgs a{};
    a.x        = reinterpret_cast<const float2*>(d_in);
    a.hist     = static_cast<const float2*>(c->d_hist.ptr);
    a.H        = static_cast<const float2*>(c->d_H.ptr);
    a.twB      = static_cast<const float2*>(c->d_twB.ptr);
    a.twC      = static_cast<const float2*>(c->d_twC.ptr);
    a.taps     = static_cast<const float*>(c->d_taps.ptr);
    a.out      = d_mag2;
    a.n_frames = (long)n_frames;
    const size_t lds = (size_t)kSLen * sizeof(float2) + 2 * 256 * sizeof(float2) + 256 * sizeof(float);
    static bool  attr_set = false;
    if (!attr_set) {
        GR4_HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(chain_fd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL(chain_fd_kernel, dim3((unsigned)n_frames), dim3(kT), lds, st, a);
    GR4_LAUNCH_CHECK();
    // carry the last 256 input samples for the next call's first frame (stream-ordered after the kernel's reads)
    GR4_HIP_TRY(hipMemcpyAsync(c->d_hist.ptr, a.x + n_frames * (size_t)kN - 256, 256 * sizeof(float2), hipMemcpyDeviceToDevice, st));
    return GR4HIP_OK;
}

void chain_fused_destroy(ChainFused* c) { delete c; }

} // namespace gr4
