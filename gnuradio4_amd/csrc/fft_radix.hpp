// fft_radix.hpp -- register-level butterflies shared by the fused chain kernel (chain_fused.hip) and the FFT block kernels (fft.hip)
#pragma once
#include <hip/hip_runtime.h>

namespace gr4 {

#if GR4_CMUL_PK // experiment (round 4, VERDICT #3 i): the complex product as two packed instructions with operand-select / negate modifiers instead of 2 mul + 2 fma
__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    typedef float cm_f2 __attribute__((ext_vector_type(2)));
    cm_f2       t, r;
    const cm_f2 x = {a.x, a.y}, w = {b.x, b.y};
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(x), "v"(w));          // {-a.y b.y, a.y b.x}
    asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(x), "v"(w), "v"(t));       // {a.x b.x, a.x b.y} + t
    return make_float2(r[0], r[1]);
}
#else
__device__ __forceinline__ float2 cmul(float2 a, float2 b) { return make_float2(fmaf(a.x, b.x, -a.y * b.y), fmaf(a.x, b.y, a.y * b.x)); }
#endif
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

// natural-order small DFTs (used by the run-time plan kernel and as the third pass of the compile-time plans)
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

template <int R>
__device__ __forceinline__ void dft_small(float2* v) {
    if constexpr (R == 2) fft2(v[0], v[1]);
    else if constexpr (R == 4) fft4(v[0], v[1], v[2], v[3]);
    else { float2(&a)[8] = *reinterpret_cast<float2(*)[8]>(v); fft8(a); }
}

// The passes of the compile-time plan 16 x 16 x R3 (N = 256 .. 4096, fft_kernels.hpp) on a frame whose first-pass inputs are already in
// registers: v[r] = frame[t + r N/16] for lane t of the frame's N/16 lanes.  buf = the frame's exchange buffer of N + N/32 float2 (one pad
// per 32: the stride-16 scatter of pass 1 is conflict-free).  On return X[j] = bin t + j N/16.  BARRIER() synchronises ALL lanes that share
// the workgroup (every frame of the group runs the same code); the caller guarantees nobody still reads buf when this starts.
// w2a/w2b = W_256^{k}, W_256^{2k} (k = t & 15), w3 = W_N^{t & 255}, w3sq = W_N^{2 (t & 255)} (used at N = 4096 only).
template <int LOG2N, typename Barrier>
__device__ __forceinline__ void fft_small_passes(float2 (&v)[16], float2* buf, int t, float2 w2a, float2 w2b, float2 w3, float2 w3sq, float2 (&X)[16], Barrier&& BARRIER) {
    constexpr int N = 1 << LOG2N, T = N / 16;
    constexpr int R3 = N / 256, B3 = R3 > 1 ? 16 / R3 : 1, NB3 = N / (R3 > 1 ? R3 : 1);
    static_assert(LOG2N >= 8 && LOG2N <= 12, "16 x 16 x R3 plans");
    auto P = [](int i) { return i + (i >> 5); };
    fft16<1>(v);
#pragma unroll
    for (int r = 0; r < 16; ++r) buf[P(16 * t + r)] = v[perm16(r)];
    BARRIER();
#pragma unroll
    for (int r = 0; r < 16; ++r) v[r] = buf[P(t + r * T)];
    BARRIER();
    apply_powers(v, w2a, w2b);
    fft16<1>(v);
    if constexpr (R3 == 1) {
#pragma unroll
        for (int q = 0; q < 16; ++q) X[q] = v[perm16(q)];
    } else {
#pragma unroll
        for (int q = 0; q < 16; ++q) buf[P((t & ~15) * 16 + (t & 15) + 16 * q)] = v[perm16(q)];
        BARRIER();
#pragma unroll
        for (int b = 0; b < B3; ++b)
#pragma unroll
            for (int r = 0; r < R3; ++r) v[b * R3 + r] = buf[P(t + b * T + r * NB3)];
        BARRIER();
        if constexpr (R3 == 16) {
            apply_powers(v, w3, w3sq);
            fft16<1>(v);
        } else {
#pragma unroll
            for (int b = 0; b < B3; ++b) {
                const float2 wb = b == 0 ? w3 : cmul(w3, w32(2 * b));
                float2       pw = wb;
#pragma unroll
                for (int r = 1; r < R3; ++r) {
                    v[b * R3 + r] = cmul(v[b * R3 + r], pw);
                    if (r + 1 < R3) pw = cmul(pw, wb);
                }
                dft_small<R3>(v + b * R3);
            }
        }
#pragma unroll
        for (int b = 0; b < B3; ++b)
#pragma unroll
            for (int q = 0; q < R3; ++q) X[b + q * B3] = v[b * R3 + (R3 == 16 ? perm16(q) : q)];
    }
}

} // namespace gr4
