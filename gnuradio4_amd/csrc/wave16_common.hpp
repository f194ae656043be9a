// wave16_common.hpp -- building blocks shared by the 8-points-per-lane kernels (chain16.hip, fir_decim_fd.hip): LDS-DMA pieces and raw barriers from
// inline asm, single-issue 8-byte LDS reads, split barriers on LDS counters, and the wave-private 512-point transform (radix 8 x 8 x 8).
#pragma once
#include "common.hpp"
#include "fft_radix.hpp"

namespace gr4 {

typedef __attribute__((address_space(3))) void* lds16_ptr_t;
__device__ __forceinline__ unsigned lds_off(const void* p) { return (unsigned)(uintptr_t)(lds16_ptr_t)p; }

// 1 KiB LDS-DMA piece from inline asm (see chain_fused.hip: hipcc must not see it as a vector-memory operation)
__device__ __forceinline__ void dma16_1k(const void* gsrc_lane, unsigned lds_byte_addr) {
    unsigned keep;
    lds_byte_addr = __builtin_amdgcn_readfirstlane(lds_byte_addr);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc_lane), "s"(lds_byte_addr)
                 : "memory");
}
#define G16_LDS_BARRIER() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define G16_FULL_BARRIER() asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define G16_FENCE()                            \
    do {                                       \
        asm volatile("" ::: "memory");         \
        __builtin_amdgcn_sched_barrier(0);     \
    } while (0)

// 8-byte LDS reads from inline asm.  hipcc pairs neighbouring ds_read_b64 into ds_read2_b64 / ds_read2st64_b64, which the LDS serves at half
// the rate of two single reads (MI355X_MICROARCH.md, LDS table: 8 cycles against 2 x 2); the kernel is LDS-bound, so every 8-byte read is issued
// by hand and the values are tied to ONE explicit s_waitcnt (lds_wait8: the "+v" operands make every use depend on the wait).
typedef float f2v __attribute__((ext_vector_type(2)));
#define G16_RD(dst, addr, off) asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
#define G16_RD8(d, addr, st)                                                                                              \
    do {                                                                                                                  \
        G16_RD(d[0], addr, 0 * (st)); G16_RD(d[1], addr, 1 * (st)); G16_RD(d[2], addr, 2 * (st)); G16_RD(d[3], addr, 3 * (st)); \
        G16_RD(d[4], addr, 4 * (st)); G16_RD(d[5], addr, 5 * (st)); G16_RD(d[6], addr, 6 * (st)); G16_RD(d[7], addr, 7 * (st)); \
    } while (0)
__device__ __forceinline__ void lds_wait8(f2v (&d)[8]) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]), "+v"(d[4]), "+v"(d[5]), "+v"(d[6]), "+v"(d[7])::"memory");
}
__device__ __forceinline__ void unpack8(float2 (&v)[8], const f2v (&d)[8]) {
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = make_float2(d[i][0], d[i][1]);
}

// split barrier on an LDS counter: arrive() after this wave's accesses in question have been ISSUED (the LDS serves a wave in order),
// wait() before the first access that must come after every wave's arrival
__device__ __forceinline__ void split_arrive(unsigned* cnt, int lane) {
    if (lane == 0) __hip_atomic_fetch_add(cnt, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__device__ __forceinline__ void split_wait(unsigned* cnt, unsigned target) {
    unsigned v;
    do {
        v = __builtin_amdgcn_readfirstlane(__hip_atomic_load(cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP));
    } while ((int)(v - target) < 0);
    asm volatile("" ::: "memory");
}

// v[k] *= W_16^k, k = 1..7
__device__ __forceinline__ void mul_w16_powers(float2 (&u)[8]) {
    constexpr float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f, hq = 0.70710678118654752440f;
    u[1] = cmul(u[1], make_float2(c1, -s1));
    u[2] = make_float2((u[2].x + u[2].y) * hq, (u[2].y - u[2].x) * hq);
    u[3] = cmul(u[3], make_float2(s1, -c1));
    u[4] = mul_mi(u[4]);
    u[5] = cmul(u[5], make_float2(-s1, -c1));
    u[6] = make_float2((u[6].y - u[6].x) * hq, (-u[6].x - u[6].y) * hq);
    u[7] = cmul(u[7], make_float2(-c1, -s1));
}

#ifndef GR4_C16_SWAP
#define GR4_C16_SWAP 0
#endif
#if GR4_C16_SWAP
// v_permlane32_swap / v_permlane16_swap (gfx950): the upper half (odd 16-lane rows) of `a` and the lower half (even rows) of `b` trade places.
// Applied to the register pairs (v[j], v[j + 4]) it exchanges an index bit that lives in the lane id (bit 5 / bit 4) with one that lives in the
// register number: a radix-2 level across lanes becomes an in-lane butterfly at one VALU instruction per dword moved, no LDS round trip.
__device__ __forceinline__ void swap32c(float2& a, float2& b) {
    const auto rx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const auto ry = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = make_float2(__uint_as_float(rx[0]), __uint_as_float(ry[0]));
    b = make_float2(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
}
__device__ __forceinline__ void swap16c(float2& a, float2& b) {
    const auto rx = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
    const auto ry = __builtin_amdgcn_permlane16_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
    a = make_float2(__uint_as_float(rx[0]), __uint_as_float(ry[0]));
    b = make_float2(__uint_as_float(rx[1]), __uint_as_float(ry[1]));
}

// The rest of the wave-private 512-point transform (decimation in frequency, 8 x 4 x 8 x 2) after its first radix-8 stage.
//   in : v[ka] = A'_l[ka] on lane l = (l5 l4 l3 l2 l1 l0)   (stage-1 output, already times W_512^{l ka})
//   1. lane bits 5, 4 <-> register bits 2, 1 (two swap rounds): registers now hold c = (l5 l4) and ka0 -> two radix-4 butterflies over c -> kb0,
//      times W_64^{l' kb0}, l' = l & 15 (twB[1..3])
//   2. ONE exchange through the private region: element (ka, kb0, l') at 17 X + l', X = ka + 8 kb0; lane X + 32 l0 reads l' = 2 d + l0, d = 0..7
//   3. radix-8 over d -> kc0, times W_16^{l0 kc0} (upper half wave), lane bit 5 <-> register bit 2, radix-2 over l0 -> kc1
//   out: v[r] = Z[k'], k' = X + 128 (lane >> 5) + 32 (r & 3) + 256 (r >> 2), X = lane & 31
__device__ __forceinline__ void private_tail(float2 (&v)[8], float2* R, int l, const float2 (&twB)[4]) {
#pragma unroll
    for (int j = 0; j < 4; ++j) swap32c(v[j], v[j + 4]);
#pragma unroll
    for (int j = 0; j < 8; ++j)
        if ((j & 2) == 0) swap16c(v[j], v[j + 2]);
    fft4(v[0], v[2], v[4], v[6]);
    fft4(v[1], v[3], v[5], v[7]);
#pragma unroll
    for (int r = 2; r < 8; ++r) v[r] = cmul(v[r], twB[r >> 1]);
    // element (ka = ka0 + 2 L4 + 4 L5, kb0 = r >> 1, l') -> 17 (ka + 8 kb0) + l'
    const int wbase = 17 * (2 * ((l >> 4) & 1) + 4 * (l >> 5)) + (l & 15);
#pragma unroll
    for (int r = 0; r < 8; ++r) R[wbase + 17 * ((r & 1) + 8 * (r >> 1))] = v[r];
    {
        f2v            d[8];
        const unsigned a1 = lds_off(R) + 8u * (unsigned)(17 * (l & 31) + (l >> 5));
        G16_RD8(d, a1, 16); // l' = 2 d + l0
        lds_wait8(d);
        unpack8(v, d);
    }
    fft8(v); // -> kc0
    if (l >= 32) mul_w16_powers(v);
#pragma unroll
    for (int j = 0; j < 4; ++j) swap32c(v[j], v[j + 4]);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float2 p = v[j], q = v[j + 4];
        v[j]     = cadd(p, q);
        v[j + 4] = csub(p, q);
    }
}
// bin k' of register r on lane l after private_tail
__host__ __device__ constexpr int c16_out_bin(int l, int r) { return (l & 31) + 128 * (l >> 5) + 32 * (r & 3) + 256 * (r >> 2); }
#else
// stages 2 and 3 of the wave-private 512-point transform (m = 64 r + l -> k' = ka + 8 kb0 + 64 kb1).  In: v[ka] = stage-1 output of lane l
// (already times W_512^{l ka}).  Out: v[kb1] = Z[ka' + 8 kb0' + 64 kb1] on lane 8 ka' + kb0'.  R = this wave's private region (float2 index
// space); rows of 72 / 65 float2 keep both transposes bank-conflict-free.
__device__ __forceinline__ void private_tail(float2 (&v)[8], float2* R, int l, const float2 (&tw2)[8]) {
    const int      hi = l >> 3, lo = l & 7;
    const unsigned rb = lds_off(R);
    f2v            d[8];
#pragma unroll
    for (int ka = 0; ka < 8; ++ka) R[72 * ka + l] = v[ka];
    {
        const unsigned a1 = rb + 8u * (unsigned)(72 * hi + lo);
        G16_RD8(d, a1, 64); // A'[8 l1 + l0][ka' = hi], l1 = 0..7
        lds_wait8(d);
        unpack8(v, d);
    }
    fft8(v);                                                          // -> kb0
#pragma unroll
    for (int k = 1; k < 8; ++k) v[k] = cmul(v[k], tw2[k]);            // W_64^{l0 kb0}
#pragma unroll
    for (int k = 0; k < 8; ++k) R[65 * k + l] = v[k];
    {
        const unsigned a2 = rb + 8u * (unsigned)(65 * lo + 8 * hi);
        G16_RD8(d, a2, 8); // C'[(ka' = hi, l0)][kb0' = lo], l0 = 0..7
        lds_wait8(d);
        unpack8(v, d);
    }
    fft8(v);                                                          // -> kb1
}
__host__ __device__ constexpr int c16_out_bin(int l, int r) { return (l >> 3) + 8 * (l & 7) + 64 * r; }
#endif

} // namespace gr4
