// fir_f16_common.hpp -- what the f16 matrix-pipe kernels share (fir_f16.hip, fir_decim_f16.hip): two- and three-term f16 splits under a block scale, wave / column / row
// reductions on the DPP network, IEEE binary16 conversions on the host
#pragma once
#include "common.hpp"

#include <cmath>
#include <cstring>

namespace gr4 {

using f16x8_h = __attribute__((ext_vector_type(8))) _Float16;
using f16x2_h = __attribute__((ext_vector_type(2))) _Float16;
using f32x4_h = __attribute__((ext_vector_type(4))) float;
using f32x2_h = __attribute__((ext_vector_type(2))) float;
using u32x4_h = __attribute__((ext_vector_type(4))) unsigned;

constexpr int kHfMaxRange = 28; // a segment whose largest sample is more than 2^28 above its ordinary level takes the float32 path

// two samples -> their two f16 terms under the block scale s (a power of two: x s is exact); the residual is exact in float32 and enters its plane times 2^11
__device__ __forceinline__ void hf_split2(float x0, float x1, float s, unsigned& h, unsigned& l) {
    const f32x2_h v  = {x0 * s, x1 * s};
    const f16x2_h hh = __builtin_convertvector(v, f16x2_h);
    // (x s - hh) 2^11 as ONE mixed-precision multiply-add per sample on the f16 term itself (v_fma_mix_f32; exact: the result is representable) -- measured + 1.5 % on the
    // decimator against converting hh back and subtracting
    const float   s2 = s * 2048.f;
    const f32x2_h r  = {__builtin_fmaf((float)hh[0], -2048.f, x0 * s2), __builtin_fmaf((float)hh[1], -2048.f, x1 * s2)};
    const f16x2_h ll = __builtin_convertvector(r, f16x2_h);
    h = __builtin_bit_cast(unsigned, hh);
    l = __builtin_bit_cast(unsigned, ll);
}

// wave-wide reductions on the DPP network (four row steps) + four scalar reads: every lane returns the wave's value.  (Six ds_bpermute steps per statistic were a
// dependent chain of ~600 cycles in front of the segment's barrier: 8 % of the kernel.)
template <typename Op>
__device__ __forceinline__ unsigned hf_wave_reduce_u32(unsigned v, Op op) {
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xF, 0xF, false));  // quad_perm [1, 0, 3, 2]
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xF, 0xF, false));  // quad_perm [2, 3, 0, 1]
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xF, 0xF, false)); // row_half_mirror
    v = op(v, (unsigned)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xF, 0xF, false)); // row_mirror: every lane holds its row of 16
    const unsigned r0 = __builtin_amdgcn_readlane((int)v, 0), r1 = __builtin_amdgcn_readlane((int)v, 16), r2 = __builtin_amdgcn_readlane((int)v, 32), r3 = __builtin_amdgcn_readlane((int)v, 48);
    return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ float hf_wave_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
    const float r0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 0)), r1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 16));
    const float r2 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 32)), r3 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 48));
    return (r0 + r1) + (r2 + r3);
}

// the sum over the four lanes (col, kq = 0 .. 3) that hold one output column, on every one of them: v_permlane16_swap / v_permlane32_swap (gfx950) of a value with itself
// put the even rows beside the odd ones / the lower half beside the upper one
__device__ __forceinline__ float hf_column_sum(float v) {
    const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    v            = __uint_as_float(a[0]) + __uint_as_float(a[1]);
    const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(b[0]) + __uint_as_float(b[1]);
}
// the smallest value of a row of 16 lanes, on every lane of the row
__device__ __forceinline__ float hf_row_min(float v) {
    v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false)));
    v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false)));
    v = __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false)));
    return __builtin_fminf(v, __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false)));
}

// the sum of a row of 16 lanes, on every lane of the row
__device__ __forceinline__ float hf_row_sum(float v) {
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0xB1, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x4E, 0xF, 0xF, false));
    v += __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x141, 0xF, 0xF, false));
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x140, 0xF, 0xF, false));
}

// the same into THREE terms (33 bits: a float32 value exactly, while the block exponent holds): the second evaluation of a segment the guard has rejected
__device__ __forceinline__ void hf_split2x3(float x0, float x1, float s, unsigned& h, unsigned& m, unsigned& l) {
    const f32x2_h v  = {x0 * s, x1 * s};
    const f16x2_h hh = __builtin_convertvector(v, f16x2_h);
    const float   s2 = s * 2048.f;
    const f32x2_h r1 = {__builtin_fmaf((float)hh[0], -2048.f, x0 * s2), __builtin_fmaf((float)hh[1], -2048.f, x1 * s2)};
    const f16x2_h mm = __builtin_convertvector(r1, f16x2_h);
    const f32x2_h r2 = {__builtin_fmaf((float)mm[0], -2048.f, r1[0] * 2048.f), __builtin_fmaf((float)mm[1], -2048.f, r1[1] * 2048.f)};
    const f16x2_h ll = __builtin_convertvector(r2, f16x2_h);
    h = __builtin_bit_cast(unsigned, hh);
    m = __builtin_bit_cast(unsigned, mm);
    l = __builtin_bit_cast(unsigned, ll);
}

static inline unsigned short host_f16_rne(float f) { // float -> IEEE binary16, round to nearest even (subnormals kept)
    unsigned u;
    std::memcpy(&u, &f, 4);
    const unsigned short sign = (unsigned short)((u >> 16) & 0x8000u);
    u &= 0x7fffffffu;
    if (u >= 0x7f800000u) return (unsigned short)(sign | (u > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (u >= 0x477ff000u) return (unsigned short)(sign | 0x7c00u); // >= 65520 rounds to infinity
    if (u < 0x38800000u) {                                         // below 2^-14: a multiple of 2^-24
        float a;
        std::memcpy(&a, &u, 4);
        return (unsigned short)(sign | (unsigned)std::nearbyint(a * 16777216.f));
    }
    unsigned       h   = (((u >> 23) - 112u) << 10) | ((u & 0x7fffffu) >> 13);
    const unsigned rem = u & 0x1fffu;
    if (rem > 0x1000u || (rem == 0x1000u && (h & 1u))) ++h;
    return (unsigned short)(sign | h);
}
static inline float host_f16_to_f(unsigned short h) {
    const int   e = (h >> 10) & 31, m = h & 1023;
    const float v = e == 0 ? std::ldexp((float)m, -24) : (e == 31 ? (m ? NAN : INFINITY) : std::ldexp((float)(1024 + m), e - 25));
    return (h & 0x8000) ? -v : v;
}

} // namespace gr4
