// fft_radix.hpp -- register-level butterflies shared by the fused chain kernel (chain_fused.hip) and the FFT block kernels (fft.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace gr4 {

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

// the three steps of fft16, callable one quarter at a time (the fused pass B interleaves them with MFMA issues)
template <int ST>
__device__ __forceinline__ void fft16_s1(float2* v, int n2) { bfly4(v[(n2)*ST], v[(n2 + 4) * ST], v[(n2 + 8) * ST], v[(n2 + 12) * ST]); }
template <int ST>
__device__ __forceinline__ void fft16_twa(float2* v) { // a[n2][k1] is at slot n2 + 4 k1; twiddle W_16^{n2 k1}
    v[5 * ST]  = mul_w16<1>(v[5 * ST]);   // n2=1,k1=1
    v[9 * ST]  = mul_w16<2>(v[9 * ST]);   // n2=1,k1=2
    v[13 * ST] = mul_w16<3>(v[13 * ST]);  // n2=1,k1=3
    v[6 * ST]  = mul_w16<2>(v[6 * ST]);   // n2=2,k1=1
    v[10 * ST] = mul_w16<4>(v[10 * ST]);  // n2=2,k1=2
}
template <int ST>
__device__ __forceinline__ void fft16_twb(float2* v) {
    v[14 * ST] = mul_w16<6>(v[14 * ST]);  // n2=2,k1=3
    v[7 * ST]  = mul_w16<3>(v[7 * ST]);   // n2=3,k1=1
    v[11 * ST] = mul_w16<6>(v[11 * ST]);  // n2=3,k1=2
    {   // n2=3,k1=3: W_16^9 = (-c1, +s1)
        constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
        const float2 a = v[15 * ST];
        v[15 * ST] = make_float2(fmaf(-a.y, s1, -a.x * c1), fmaf(a.x, s1, -a.y * c1));
    }
}
template <int ST>
__device__ __forceinline__ void fft16_s2(float2* v, int k1) { bfly4(v[(4 * k1) * ST], v[(4 * k1 + 1) * ST], v[(4 * k1 + 2) * ST], v[(4 * k1 + 3) * ST]); }

template <int ST>
__device__ __forceinline__ void fft16(float2* v) {
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) fft16_s1<ST>(v, n2);
    fft16_twa<ST>(v);
    fft16_twb<ST>(v);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) fft16_s2<ST>(v, k1);
}

// multiply v[r] by b^r (r = 1..15) given the table values b^1 and b^2
__device__ __forceinline__ void apply_powers(float2 (&v)[16], float2 b1, float2 b2) {
    // two interleaved chains (odd / even powers) stepping by b^2: 14 multiplies, only three powers live at any time
    float2 wo = b1, we = b2;
    v[1] = cmul(v[1], wo);
    v[2] = cmul(v[2], we);
#pragma unroll
    for (int r = 3; r < 16; r += 2) {
        wo   = cmul(wo, b2);
        v[r] = cmul(v[r], wo);
        if (r + 1 < 16) {
            we       = cmul(we, b2);
            v[r + 1] = cmul(v[r + 1], we);
        }
    }
}

// value of the neighbouring lane (lane ^ 1) through DPP quad_perm [1,0,3,2]: no LDS traffic
__device__ __forceinline__ float lane_xor1(float v) {
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
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

} // namespace gr4
