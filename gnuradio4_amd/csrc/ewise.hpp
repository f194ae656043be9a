// ewise.hpp -- element-wise programs: a run of adjacent per-sample blocks executed on values in registers.
//
// The reference fuses adjacent blocks at compile time: Merge<A, "out", B, "in"> is ONE processOne() that calls A's and then B's on the value it holds, "bypassing
// runtime buffers" (core/include/gnuradio-4.0/BlockMerging.hpp:126-240); its only published table measures exactly that (docs/USER_API_Connecting_Blocks.md:207-222:
// mult -> div -> add and (mult -> div -> add)^10).  The run-time counterpart here: the per-sample blocks of the hot path -- MathOpImpl<T, op> (blocks/math/.../Math.hpp:38-56)
// and Rotator<complex<float>> (Rotator.hpp:51-61, closed-form phase) -- are OPS of a program; a run of them is one list in device memory, and a kernel applies the
// whole list to each value between its load and its store.  Three users:
//   * ewise.hip: the program alone -- one launch, 2 sizeof(T) bytes of HBM traffic per sample whatever the length of the chain;
//   * load hooks (prologue) and store hooks (epilogue) of the FIR kernels: the neighbours of a filter ride in ITS launch.
// Semantics are the blocks': integer ops promote, wrap and narrow like C++ (bit-exact), float ops are single IEEE operations in program order (never contracted
// into fused multiply-adds), a rotator op multiplies by exp(j (phase0 + (k + 1) inc)) with k the sample's ABSOLUTE index in the stream (the position is an
// argument of apply, so a program carries no mutable state).
//
// What the device walks is a list of ITEMS compiled from the ops on the host (ewise.hip), each bit-identical to the ops it stands for:
//   real sample types:  e = ((e * m) + a) / c   -- m: a MultiplyConst (or 1), a: an AddConst / SubtractConst behind it (or -0.0: x + -0.0 == x bit for bit), c: a
//                       DivideConst behind that (flag; a float quotient by a power of two is the product with its reciprocal, bit-identical, and becomes an m).
//                       Floats: one item per mult / add pair, each operation rounded by itself.  Integers: +, -, * modulo 2^w form a ring, so a whole run of
//                       them -- any length -- collapses into ONE item (M, A); only truncating divisions separate items.
//                       Items without a division run in a branch-free loop (one multiply and one add per value and item whatever the item says: the
//                       multi-way branch on the op kind cost a register copy per value and arm).
//   complex types:      one item per op (add / subtract / multiply / divide by a complex constant, rotate), dispatched by a uniform branch.
#pragma once
#include "common.hpp"

namespace gr4 {

enum : int { kEwAdd = GR4HIP_ADD, kEwSub = GR4HIP_SUB, kEwMul = GR4HIP_MUL, kEwDiv = GR4HIP_DIV, kEwRotate = 4, kEwAffine = 5 };
enum : int { kEwFlagDiv = 1, kEwFlagDivRcp = 2, kEwFlagNan = 4 }; // kEwFlagDivRcp (float): the correctly rounded reciprocal of c sits behind c (raw + 20)

struct alignas(16) EwiseOp { // 32 bytes
    int kind, flags;
    union {
        unsigned char raw[24]; // kEwAffine: m at 0, a at 8, c at 16 (each one T); complex ops: the constant at 0
        unsigned long long q[3]; // kEwRotate: {phase0, increment} as 64-bit fractions of a turn (common.hpp: turns_fix); flags = kEwFlagNan: a non-finite phase or increment
    } u;
};
// programs are read through the scalar cache (uniform address, constant address space): one s_load per item and wave, no vector registers
typedef const __attribute__((address_space(4))) EwiseOp* EwiseProg;
inline EwiseProg as_prog(const void* d_ops) { return (EwiseProg)(reinterpret_cast<uintptr_t>(d_ops)); }

__device__ __forceinline__ EwiseOp ew_load(EwiseProg ops, int k) { // four scalar 64-bit words (a class copy out of the constant address space does not bind)
    const auto* q = (const __attribute__((address_space(4))) unsigned long long*)ops + 4 * (long)k;
    const unsigned long long w[4] = {q[0], q[1], q[2], q[3]};
    EwiseOp o;
    __builtin_memcpy(&o, w, sizeof(o));
    return o;
}

struct EwiseHook { // a program attached to another kernel's load or store: pos = absolute stream index of the kernel's sample 0
    EwiseProg ops     = nullptr;
    int       n_ops   = 0;
    int       has_div = 0; // some item of a real-typed program carries a division
    long      pos     = 0;
    // the program is ONE finite rotator on complex<float> (a down-converter's mixer: the channeliser's front end): its constants ride in the hook itself, and a kernel
    // that knows the order in which it touches samples steps the phase by integer additions (bit-identical to walking the program: the phase is exact modulo 2^64)
    int                rotor_only = 0;
    unsigned long long rot_p0 = 0, rot_inc = 0;
};

template <typename T> struct EwWide { using type = uint32_t; }; // +, -, * modulo 2^32 (then narrowed): what promotion to int, the operation and the narrowing give
template <> struct EwWide<uint64_t> { using type = uint64_t; };
template <> struct EwWide<int64_t> { using type = uint64_t; };
template <> struct EwWide<float> { using type = float; };
template <> struct EwWide<double> { using type = double; };

template <typename T> struct ew_is_complex : std::false_type {};
template <> struct ew_is_complex<float2> : std::true_type {};
template <> struct ew_is_complex<double2> : std::true_type {};

template <typename T>
__device__ __forceinline__ T ew_div(T a, T b) {
    if constexpr (std::is_integral_v<T>) return b == T(0) ? T(0) : (T)(a / b); // x / 0 is UB in the reference; defined as 0 here (as gr4hip_math_const)
    else return a / b;
}

// x / c with y = RN(1 / c) known (Markstein: q = RN(x y), r = x - q c exactly by one fma, q' = RN(q + r y) is the correctly rounded quotient when y is the correctly
// rounded reciprocal).  Taken only where nothing under- or overflows on the way: |q| inside [2^-60, 2^60] with |c| inside [2^-40, 2^40] (the host checks c; c's
// significand must not be all ones); everything else -- zeros, infinities, NaNs, the extremes -- takes the general quotient.  tests/test_gpu_fusion.py compares the
// two over ALL 2^32 float inputs for a set of constants.
__device__ __forceinline__ float ew_div_by_const(float x, float c, float y) {
#pragma clang fp contract(off)
    const float q  = x * y;
    const float r  = __builtin_fmaf(-q, c, x);
    const float q2 = __builtin_fmaf(r, y, q);
    const float aq = __builtin_fabsf(q);
    if (aq >= 0x1p-60f && aq <= 0x1p60f) return q2;
    return x / c;
}

template <typename T, int KIND>
__device__ __forceinline__ T ew_cconst(T a, T b) { // std::complex arithmetic on interleaved pairs
#pragma clang fp contract(off)
    T r;
    if constexpr (KIND == kEwAdd) { r.x = a.x + b.x; r.y = a.y + b.y; }
    else if constexpr (KIND == kEwSub) { r.x = a.x - b.x; r.y = a.y - b.y; }
    else if constexpr (KIND == kEwMul) { r.x = a.x * b.x - a.y * b.y; r.y = a.x * b.y + a.y * b.x; }
    else {
        const auto d = b.x * b.x + b.y * b.y;
        r.x = (a.x * b.x + a.y * b.y) / d;
        r.y = (a.y * b.x - a.x * b.y) / d;
    }
    return r;
}

// exp(j 2 pi (ph0 + k inc)) with k = absolute sample index + 1, the arithmetic of rotator_closed_kernel (math.hip): the phase as a 64-bit fraction of a turn, exact
// modulo one turn (common.hpp: turns_fix, rotor_at).  A rotator as the load hook of a matrix-pipe decimator is bound by exactly this: one 64-bit product per CALL (the
// first element), two integer additions per further element (their index differences are compile-time constants in every kernel: the products are scalar).
__device__ __forceinline__ unsigned long long ew_rotor_phase(const EwiseOp& op, long k) { return op.u.q[0] + (unsigned long long)k * op.u.q[1]; }
__device__ __forceinline__ void ew_rotor(const EwiseOp& op, unsigned long long phase, float& cs, float& sn) {
#ifdef GR4_T_ROTOR_TRIVIAL // timing-only build: what a hooked launch costs without the rotor's arithmetic (wrong results)
    cs = __builtin_bit_cast(float, (int)phase & 0x3f800000);
    sn = (float)op.flags;
    return;
#endif
    rotor_at(phase, cs, sn);
    if (op.flags & kEwFlagNan) cs = sn = __builtin_nanf(""); // (uniform)
}

// apply the whole program to NE values held by this lane; index(j) = absolute stream index of element j (rotator ops only)
template <typename T, int NE, typename IndexFn>
__device__ __forceinline__ void ewise_apply(T (&e)[NE], EwiseProg ops, int n_ops, int has_div, IndexFn&& index) {
    if (n_ops <= 0) return;
    EwiseOp cur = ew_load(ops, 0);
    if constexpr (!ew_is_complex<T>::value) {
        using W = typename EwWide<T>::type;
        const auto mul_add = [&](const EwiseOp& op) {
#pragma clang fp contract(off) // one IEEE operation per source operation: what the blocks on the host compute
            T m, a;
            __builtin_memcpy(&m, op.u.raw, sizeof(T));
            __builtin_memcpy(&a, op.u.raw + 8, sizeof(T));
            if constexpr (std::is_same_v<T, float> && NE % 2 == 0) { // two values per instruction: v_pk_mul_f32 / v_pk_add_f32 (IEEE per component like the scalar forms)
                typedef float ew_f32x2 __attribute__((ext_vector_type(2)));
                const ew_f32x2 m2 = {m, m}, a2 = {a, a};
#pragma unroll
                for (int j = 0; j < NE; j += 2) {
                    ew_f32x2 v = {e[j], e[j + 1]};
                    v          = v * m2;
                    v          = v + a2;
                    e[j]       = v[0];
                    e[j + 1]   = v[1];
                }
            } else {
#pragma unroll
                for (int j = 0; j < NE; ++j) {
                    const W p = (W)e[j] * (W)m;
                    e[j]      = (T)(p + (W)a);
                }
            }
        };
        if (!has_div) { // the common case: no branch inside the loop
            for (int k = 0; k < n_ops; ++k) {
                const EwiseOp nxt = ew_load(ops, k + 1 < n_ops ? k + 1 : k); // requested before this item's arithmetic: its latency hides behind 2 NE operations
                mul_add(cur);
                cur = nxt;
            }
        } else {
            for (int k = 0; k < n_ops; ++k) {
                const EwiseOp nxt = ew_load(ops, k + 1 < n_ops ? k + 1 : k);
                mul_add(cur);
                if (cur.flags & kEwFlagDiv) {
                    T c;
                    __builtin_memcpy(&c, cur.u.raw + 16, sizeof(T));
                    if constexpr (std::is_same_v<T, float>) {
                        if (cur.flags & kEwFlagDivRcp) { // x / c for a constant c: three operations instead of the ~11 of the general quotient, the same bits (below)
                            float y;
                            __builtin_memcpy(&y, cur.u.raw + 20, 4);
#pragma unroll
                            for (int j = 0; j < NE; ++j) e[j] = ew_div_by_const(e[j], c, y);
                            cur = nxt;
                            continue;
                        }
                    }
#pragma unroll
                    for (int j = 0; j < NE; ++j) e[j] = ew_div<T>(e[j], c);
                }
                cur = nxt;
            }
        }
    } else {
        for (int k = 0; k < n_ops; ++k) {
            const EwiseOp nxt = ew_load(ops, k + 1 < n_ops ? k + 1 : k);
            T v;
            __builtin_memcpy(&v, cur.u.raw, sizeof(T));
            switch (cur.kind) { // uniform: a scalar branch
            case kEwAdd:
#pragma unroll
                for (int j = 0; j < NE; ++j) e[j] = ew_cconst<T, kEwAdd>(e[j], v);
                break;
            case kEwSub:
#pragma unroll
                for (int j = 0; j < NE; ++j) e[j] = ew_cconst<T, kEwSub>(e[j], v);
                break;
            case kEwMul:
#pragma unroll
                for (int j = 0; j < NE; ++j) e[j] = ew_cconst<T, kEwMul>(e[j], v);
                break;
            case kEwDiv:
#pragma unroll
                for (int j = 0; j < NE; ++j) e[j] = ew_cconst<T, kEwDiv>(e[j], v);
                break;
            default:
                if constexpr (std::is_same_v<T, float2>) {
                    const long               i0  = index(0);
                    const unsigned long long ph0 = ew_rotor_phase(cur, i0 + 1);
#pragma unroll
                    for (int j = 0; j < NE; ++j) {
#pragma clang fp contract(off)
                        float cs, sn;
                        ew_rotor(cur, ph0 + (unsigned long long)(index(j) - i0) * cur.u.q[1], cs, sn);
                        const float2 x = e[j];
                        e[j] = make_float2(x.x * cs - x.y * sn, x.x * sn + x.y * cs);
                    }
                }
                break;
            }
            cur = nxt;
        }
    }
}

// NE consecutive samples through a hook: sample j has index i0 + j relative to the kernel's sample 0
template <typename T, int NE>
__device__ __forceinline__ void ewise_hook(T (&e)[NE], const EwiseHook& h, long i0) {
    ewise_apply<T, NE>(e, h.ops, h.n_ops, h.has_div, [&](int j) { return h.pos + i0 + j; });
}
template <typename T>
__device__ __forceinline__ T ewise_hook1(T x, const EwiseHook& h, long i) {
    T e[1] = {x};
    ewise_hook<T, 1>(e, h, i);
    return e[0];
}

} // namespace gr4

// ---- host side (ewise.hip): the program object behind gr4hip_ewise_t, shared with the kernels that take hooks
struct gr4hip_ewise {
    int                       dtype = GR4HIP_F32;
    std::vector<gr4::EwiseOp> user;    // the ops as appended (what the chain of blocks says): kind = gr4hip_op | kEwRotate, the constant at u.raw
    std::vector<gr4::EwiseOp> ops;     // the items the device walks, compiled from `user` before the first launch after a change
    int                       has_div = 0;
    gr4::DeviceBuffer         d_ops;
    bool                      dirty = true;
    long                      pos   = 0; // samples processed since create / reset: the absolute index a rotator op's phase is a function of
};
namespace gr4 {
int           ewise_device_ops(gr4hip_ewise* p, EwiseHook* hook, hipStream_t st); // uploads on first use after a change, on `st` (the stream of the launch that will run it); hook->pos = p->pos
bool          ewise_as_real_gain(const gr4hip_ewise* p, double* gain);
gr4hip_ewise* ewise_clone(const gr4hip_ewise* p);
int           ewise_run(const EwiseHook& prog, int dtype, const void* in, void* out, long n, hipStream_t st);
} // namespace gr4
