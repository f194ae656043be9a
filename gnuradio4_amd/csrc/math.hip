// math.hip -- blocks/math kernels (MathOpImpl, MathOpMultiPortImpl, Rotator) and the synthetic-input generator.
//
// MathOpImpl<T,op>::processOne (blocks/math/.../Math.hpp:38-56) and MathOpMultiPortImpl::processBulk (:100-107) are
// pure streaming element-wise work: HBM-bound, 16-byte vector accesses, grid-stride, all n_inputs streams folded in
// ONE pass (the reference makes n-1 passes over the output).  Integer types keep C++ semantics: operands promoted,
// result narrowed back to T (wrap-around) -- bit-exact against the oracle.
#include "common.hpp"

#include <cstdlib>

namespace gr4 {

template <typename T> struct Wide { using type = T; };
template <> struct Wide<uint8_t> { using type = int; };   // integral promotion (Math.hpp:54: op()(a, value) on T operands)
template <> struct Wide<uint16_t> { using type = int; };
template <> struct Wide<int8_t> { using type = int; };
template <> struct Wide<int16_t> { using type = int; };
template <> struct Wide<int32_t> { using type = uint32_t; }; // wrap-around without signed-overflow UB (+,-,*)
template <> struct Wide<int64_t> { using type = uint64_t; };

template <typename T, int OP>
__device__ __forceinline__ T apply_op(T a, T b) {
    if constexpr (OP == GR4HIP_DIV) {
        if constexpr (std::is_integral_v<T>) return b == T(0) ? T(0) : (T)(a / b); // x/0 is UB in the reference; defined as 0 here
        else return a / b;
    } else {
        using W = typename Wide<T>::type;
        const W x = (W)a, y = (W)b;
        if constexpr (OP == GR4HIP_ADD) return (T)(x + y);
        else if constexpr (OP == GR4HIP_SUB) return (T)(x - y);
        else return (T)(x * y);
    }
}

// std::complex<T> arithmetic on interleaved pairs
template <typename F2, int OP>
__device__ __forceinline__ F2 apply_cop(F2 a, F2 b) {
    F2 r;
    if constexpr (OP == GR4HIP_ADD) { r.x = a.x + b.x; r.y = a.y + b.y; }
    else if constexpr (OP == GR4HIP_SUB) { r.x = a.x - b.x; r.y = a.y - b.y; }
    else if constexpr (OP == GR4HIP_MUL) { r.x = a.x * b.x - a.y * b.y; r.y = a.x * b.y + a.y * b.x; }
    else {
        const auto d = b.x * b.x + b.y * b.y;
        r.x = (a.x * b.x + a.y * b.y) / d;
        r.y = (a.y * b.x - a.x * b.y) / d;
    }
    return r;
}

// gr::UncertainValue<float | double> (meta/.../UncertainValue.hpp:34-40): {value, uncertainty}; both operands uncertain, real value types: the reference's operators
// (:121-250) propagate uncorrelated errors -- one IEEE operation per source operation on the value, std::hypot for the combination
template <typename F> struct UVal { F v, u; };
__device__ __forceinline__ float  u_hypot(float a, float b) { return hypotf(a, b); }
__device__ __forceinline__ double u_hypot(double a, double b) { return hypot(a, b); }
template <typename F, int OP>
__device__ __forceinline__ UVal<F> apply_uop(UVal<F> a, UVal<F> b) {
#pragma clang fp contract(off)
    if constexpr (OP == GR4HIP_ADD) return {a.v + b.v, u_hypot(a.u, b.u)};                                  // :121-133
    else if constexpr (OP == GR4HIP_SUB) return {a.v - b.v, u_hypot(a.u, b.u)};                             // :159-171
    else if constexpr (OP == GR4HIP_MUL) return {a.v * b.v, u_hypot(a.v * b.u, b.v * a.u)};                 // :192-204
    else return {a.v / b.v, u_hypot(a.u / b.v, b.u * a.v / (b.v * b.v))};                                   // :221-243
}

template <typename T, int OP>
__device__ __forceinline__ T apply_any(T a, T b) {
    if constexpr (std::is_same_v<T, float2> || std::is_same_v<T, double2>) return apply_cop<T, OP>(a, b);
    else if constexpr (std::is_same_v<T, UVal<float>>) return apply_uop<float, OP>(a, b);
    else if constexpr (std::is_same_v<T, UVal<double>>) return apply_uop<double, OP>(a, b);
    else return apply_op<T, OP>(a, b);
}

constexpr int kMaxInputs = 32; // Math.hpp:90 Limits<1U, 32U>
struct NaryPtrs { const void* p[kMaxInputs]; };

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
template <typename T> union Vec16 { u32x4 u; T e[16 / sizeof(T)]; };
// Launch shape of the streaming kernels (tools/ubench/stream_rate.hip, 2^28 int32 in + out): a workgroup owns ONE contiguous 16 KiB slab
// (4 x 256 lanes x 16 B) and the accesses are non-temporal -- 5.9 TB/s; a persistent grid-stride loop over 2048 workgroups with the
// default cache policy reaches 4.8 (hipMemcpyDtoD: 5.2).
constexpr int kMathSlab = 4; // 16-byte vectors per lane

// out[i] = ((in0[i] op in1[i]) op in2[i]) ...   or, with CONST, out[i] = in0[i] op value
template <typename T, int OP, bool CONST>
__global__ void math_kernel(NaryPtrs ins, int n_inputs, T value, T* __restrict__ out, long n, long head, long nvec) {
    // elements [head, head + nvec VE) are 16-byte aligned in every stream and go through the vector body; [0, head) and the rest take the scalar loop
    // (a ring span may start at any element: head = the elements up to the next 16-byte boundary when all streams share one misalignment, else nvec = 0)
    constexpr int VE     = 16 / sizeof(T);
    const long    stride = (long)gridDim.x * blockDim.x;
    const long    v0     = (long)blockIdx.x * (blockDim.x * kMathSlab) + threadIdx.x;
#pragma unroll
    for (int s = 0; s < kMathSlab; ++s) {
        const long v = v0 + (long)s * blockDim.x;
        if (v >= nvec) break;
        Vec16<T> acc;
        acc.u = __builtin_nontemporal_load(&reinterpret_cast<const u32x4*>(static_cast<const T*>(ins.p[0]) + head)[v]);
        if constexpr (CONST) {
#pragma unroll
            for (int e = 0; e < VE; ++e) acc.e[e] = apply_any<T, OP>(acc.e[e], value);
        } else {
            for (int k = 1; k < n_inputs; ++k) {
                Vec16<T> b;
                b.u = __builtin_nontemporal_load(&reinterpret_cast<const u32x4*>(static_cast<const T*>(ins.p[k]) + head)[v]);
#pragma unroll
                for (int e = 0; e < VE; ++e) acc.e[e] = apply_any<T, OP>(acc.e[e], b.e[e]);
            }
        }
        __builtin_nontemporal_store(acc.u, &reinterpret_cast<u32x4*>(out + head)[v]);
    }
    const long body_end = head + nvec * VE, nscalar = head + (n - body_end);
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < nscalar; j += stride) { // head and tail elements
        const long i = j < head ? j : body_end + (j - head);
        T a = static_cast<const T*>(ins.p[0])[i];
        if constexpr (CONST) a = apply_any<T, OP>(a, value);
        else
            for (int k = 1; k < n_inputs; ++k) a = apply_any<T, OP>(a, static_cast<const T*>(ins.p[k])[i]);
        out[i] = a;
    }
}

template <typename T, bool CONST>
static int math_dispatch_op(int op, const NaryPtrs& ins, int n_inputs, T value, void* out, long n, hipStream_t st) {
    constexpr long VE = 16 / sizeof(T);
    // common misalignment of all streams (in elements up to the next 16-byte boundary), or the all-scalar path when they differ
    const auto mis = [](const void* p) { return (long)((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15); };
    long head = mis(out);
    bool same = head % (long)sizeof(T) == 0;
    for (int k = 0; k < n_inputs; ++k) same = same && mis(ins.p[k]) == head;
    head = same ? std::min<long>(head / (long)sizeof(T), n) : 0;
    const long nvec = same ? (n - head) / VE : 0;
    GR4_REQUIRE(ceil_div(nvec + 1, 256L * kMathSlab) < (1L << 31), "math: span too long for one launch");
    const long     nscalar = n - nvec * VE; // head + tail, or everything when the streams are misaligned against each other
    const unsigned grid = (unsigned)std::max<long>({ceil_div(nvec + 1, 256L * kMathSlab), std::min<long>(ceil_div(nscalar, 1024L), 16384L), 1L}); // one 16 KiB slab per workgroup; the scalar loop strides over the grid
    T*             o    = static_cast<T*>(out);
    switch (op) {
    case GR4HIP_ADD: hipLaunchKernelGGL((math_kernel<T, GR4HIP_ADD, CONST>), dim3(grid), dim3(256), 0, st, ins, n_inputs, value, o, n, head, nvec); break;
    case GR4HIP_SUB: hipLaunchKernelGGL((math_kernel<T, GR4HIP_SUB, CONST>), dim3(grid), dim3(256), 0, st, ins, n_inputs, value, o, n, head, nvec); break;
    case GR4HIP_MUL: hipLaunchKernelGGL((math_kernel<T, GR4HIP_MUL, CONST>), dim3(grid), dim3(256), 0, st, ins, n_inputs, value, o, n, head, nvec); break;
    case GR4HIP_DIV: hipLaunchKernelGGL((math_kernel<T, GR4HIP_DIV, CONST>), dim3(grid), dim3(256), 0, st, ins, n_inputs, value, o, n, head, nvec); break;
    default: set_error("math: unknown op %d", op); return GR4HIP_INVALID_ARGUMENT;
    }
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

template <bool CONST>
static int math_dispatch(int op, int dtype, const NaryPtrs& ins, int n_inputs, const void* h_value, void* out, long n, hipStream_t st) {
#define GR4_CASE(ID, T)                                                              \
    case ID: {                                                                       \
        T v{};                                                                       \
        if (CONST) memcpy(&v, h_value, sizeof(T));                                   \
        return math_dispatch_op<T, CONST>(op, ins, n_inputs, v, out, n, st);         \
    }
    switch (dtype) {
        GR4_CASE(GR4HIP_U8, uint8_t)
        GR4_CASE(GR4HIP_U16, uint16_t)
        GR4_CASE(GR4HIP_U32, uint32_t)
        GR4_CASE(GR4HIP_U64, uint64_t)
        GR4_CASE(GR4HIP_I8, int8_t)
        GR4_CASE(GR4HIP_I16, int16_t)
        GR4_CASE(GR4HIP_I32, int32_t)
        GR4_CASE(GR4HIP_I64, int64_t)
        GR4_CASE(GR4HIP_F32, float)
        GR4_CASE(GR4HIP_F64, double)
        GR4_CASE(GR4HIP_C32, float2)
        GR4_CASE(GR4HIP_C64, double2)
        GR4_CASE(GR4HIP_UF32, UVal<float>)
        GR4_CASE(GR4HIP_UF64, UVal<double>)
    default: set_error("math: unknown dtype %d", dtype); return GR4HIP_INVALID_ARGUMENT;
    }
#undef GR4_CASE
}

// ------------------------------------------------------------------------------------------------ rotator
// Rotator.hpp:51-61.  The float phase recurrence is inherently sequential and is reproduced exactly:
// one lane walks the recurrence and leaves a checkpoint every kRotChunk samples; the data pass then replays
// kRotChunk steps per lane into LDS and applies cos/sin with fully coalesced 8-byte accesses.
constexpr int   kRotChunk     = 32;
constexpr float kRotLeapBelow = 0.015f; // |phase_increment| below which the leaping walker is used: one of its segment steps costs ~20 steps of the plain walker
                                        // (~150 dependent single-wave instructions against 3), and an increment of 0.015 rad averages ~20 samples per segment

__device__ __forceinline__ float rot_step(float ph, float inc) { // same values as the if / else-if of Rotator.hpp:52-58, without branches
    const float two_pi = 2.0f * 3.14159265358979323846f;
    ph += inc;
    const float lo = ph - two_pi, hi = ph + two_pi;
    return ph > two_pi ? lo : (ph < 0.0f ? hi : ph);
}

// One-sided steps: with 0 <= ph <= 2 pi before the step, a non-negative increment can only trip the upper wrap and a negative one only the
// lower wrap, so the other compare / select drops out of the dependent chain (identical values; 3 dependent VALU levels instead of 4).
template <int SIGN> // +1: inc >= 0, -1: inc < 0, 0: general
__device__ __forceinline__ float rot_step_s(float ph, float inc) {
    const float two_pi = 2.0f * 3.14159265358979323846f;
    if constexpr (SIGN == 0) return rot_step(ph, inc);
    ph += inc;
    if constexpr (SIGN > 0) return ph > two_pi ? ph - two_pi : ph;
    else return ph < 0.0f ? ph + two_pi : ph;
}

// The walker: ONE lane, the whole dependent chain.  Straight-line 32-step bodies with one checkpoint store in front of each; a per-sample
// `if (i % 32 == 0) store` loop with branches ran 7x slower.
template <int SIGN>
__global__ void rotator_checkpoint_kernel(float* __restrict__ state, float inc, float* __restrict__ ckpt, long n) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const float two_pi = 2.0f * 3.14159265358979323846f;
    float       ph    = *state;
    long        i     = 0;
    const long  nfull = n / kRotChunk;
    for (long c = 0; c < nfull; ++c) {
        ckpt[c] = ph;
        if (SIGN != 0 && ph >= 0.0f && ph <= two_pi) { // the invariant of the one-sided form holds (it is closed under the step)
#pragma unroll
            for (int k = 0; k < kRotChunk; ++k) ph = rot_step_s<SIGN>(ph, inc);
        } else {
#pragma unroll
            for (int k = 0; k < kRotChunk; ++k) ph = rot_step(ph, inc);
        }
    }
    i = nfull * kRotChunk;
    if (i < n) {
        ckpt[nfull] = ph;
        for (; i < n; ++i) ph = rot_step(ph, inc);
    }
    *state = ph;
}


// The leaping walker.  The recurrence is sequential in floating point, but not everywhere: while the phase stays inside one binade
// [2^e, 2^(e+1)) every state is a multiple of that binade's ulp u, so fl(ph + inc) = ph + c with the CONSTANT c = inc rounded to a multiple of u
// (round-to-nearest; an increment exactly half-way between two multiples would alternate with the parity of ph -- such steps are taken one at
// a time).  Inside a binade the states are an exact arithmetic progression, ph_j = ph + j c, evaluated in integers on the 24-bit mantissa; only
// the steps that change the binade or wrap at 2 pi are genuine float operations.  A 0.01 rad increment meets ~10 binades per turn of 628
// samples: ~10 sequential segment steps instead of 628 dependent adds, bit-identical states (tests: against the float recurrence of the
// oracle, 2^20 samples per increment).  One wave: all lanes carry the same walker state (uniform control flow), the lanes only share the
// checkpoint stores of a segment.
__global__ __launch_bounds__(64) void rotator_checkpoint_leap_kernel(float* __restrict__ state, float inc, float* __restrict__ ckpt, long n) {
    const float two_pi = 2.0f * 3.14159265358979323846f;
    const int   lane   = threadIdx.x;
    const unsigned Tm  = (__float_as_uint(two_pi) & 0x7fffffu) | 0x800000u; // mantissa of 2 pi in units of the ulp of [4, 8)
    float       s = *state;
    long        i = 0; // s is the state after i steps
    while (i < n) {
        const float    t    = rot_step(s, inc);
        const float    raw  = s + inc;
        const unsigned sb = __float_as_uint(s), tb = __float_as_uint(t);
        const int      es = (int)((sb >> 23) & 0xff), et = (int)((tb >> 23) & 0xff);
        long           k    = 1; // states i .. i + k - 1 are s, s + c, ...; the walker moves to state i + k
        int            C    = 0;
        const int      S    = (int)((sb & 0x7fffffu) | 0x800000u);
        const int      sh   = es - 150; // s = S * 2^sh
        bool           ap   = false;
        if (t == raw && !(sb >> 31) && es == et && es > 0 && es < 255) { // no wrap, same binade, positive normal: c = t - s is exact
            const float c = t - s, u = __builtin_amdgcn_ldexpf(1.0f, sh);
            const float r = inc - c; // exact: |inc - c| <= u / 2 and both are multiples of ulp(inc) or inc is below u
            C = (int)__builtin_amdgcn_ldexpf(c, -sh);
            if (fabsf(r) != 0.5f * u) { // not a tie: c is the step for as long as the phase stays in this binade
                // K = steps for which the progression provably equals the recurrence; all operands are below 2^24, so the quotients come from one float
                // division each (exact operands, correctly rounded quotient, fixed up by one) instead of 64-bit integer divisions
                const auto idiv = [](int num, int den) { // floor(num / den), num >= 0, den > 0, both < 2^24
                    int q = (int)((float)num / (float)den);
                    if (q * den > num) --q;
                    return q;
                };
                long K;
                if (C > 0) {
                    int Ki = idiv(0xffffff - S, C);                    // S + K C <= 2^24 - 1: below the next binade
                    if (es == 129) { const int K2 = S <= (int)Tm ? idiv((int)Tm - S, C) : 0; Ki = K2 < Ki ? K2 : Ki; } // ... and not above 2 pi (no wrap)
                    K = Ki;
                } else if (C < 0) {
                    K = S > 0x800001 ? idiv(S - 0x800001, -C) : 0;     // S + K C >= 2^23 + 1: strictly inside the binade
                } else {
                    K = n;                                             // the increment is below half an ulp here: the phase no longer moves
                }
                if (K >= 1) { ap = true; k = K < n - i ? K : n - i; }
            }
        }
        // checkpoints: state index 32 m for every m with i <= 32 m < i + k
        const long m0 = (i + kRotChunk - 1) / kRotChunk, m1 = (i + k + kRotChunk - 1) / kRotChunk; // [m0, m1)
        for (long m = m0 + lane; m < m1; m += 64) {
            const long j = m * kRotChunk - i;
            ckpt[m] = ap ? __builtin_amdgcn_ldexpf((float)(S + (int)j * C), sh) : s; // !ap: k == 1, j == 0
        }
        s = ap ? __builtin_amdgcn_ldexpf((float)(S + (int)k * C), sh) : t;
        i += k;
    }
    if (lane == 0) *state = s;
}

__global__ __launch_bounds__(256) void rotator_apply_kernel(const float2* __restrict__ x, float2* __restrict__ y, const float* __restrict__ ckpt, float inc, long n) {
    __shared__ float ph[256 * (kRotChunk + 1)];
    const long       base  = (long)blockIdx.x * 256 * kRotChunk;
    const long       chunk = (long)blockIdx.x * 256 + threadIdx.x;
    if (chunk * kRotChunk < n) {
        float p = ckpt[chunk];
#pragma unroll 8
        for (int i = 0; i < kRotChunk; ++i) {
            p = rot_step(p, inc);
            ph[threadIdx.x * (kRotChunk + 1) + i] = p;
        }
    }
    __syncthreads();
    for (int s = threadIdx.x; s < 256 * kRotChunk; s += 256) {
        const long i = base + s;
        if (i >= n) break;
        const float p = ph[(s / kRotChunk) * (kRotChunk + 1) + (s % kRotChunk)];
        float       sn, cs;
        sincosf(p, &sn, &cs);
        const float2 v = x[i];
        y[i] = make_float2(v.x * cs - v.y * sn, v.x * sn + v.y * cs);
    }
}

// Closed-form phase (GR4HIP_ROTATOR_CLOSED_FORM): sample i of a call sees phase = carried + (i + 1) inc, with phases and the increment as 64-bit fractions of a
// turn (common.hpp: turns_fix): the sum modulo 2^64 IS the reduction modulo one turn, exact however long the span (round 5; until then float64 turns with the
// product split at 2^20: four float64 operations and three conversions per sample where this takes three integer ones).  One pass, 16 B per sample: HBM-bound.
// The float recurrence of the reference drifts ~1e-7 rad per step away from this value (the walker above reproduces that drift bit for bit); this one is the
// float64 oracle's phase.
template <bool V4> // V4: 16-byte accesses (both spans 16-byte aligned); otherwise one 8-byte sample per lane (a ring span may start at any element)
__global__ __launch_bounds__(256) void rotator_closed_kernel(const float4* __restrict__ x, float4* __restrict__ y, unsigned long long ph0 /*carried phase: a 64-bit fraction of a turn, kept by the handle*/,
                                                             unsigned long long inc, int bad /*a non-finite carried phase: NaN out*/, long n) {
    const long   pairs  = (n + 1) / 2;
    auto         rot    = [&](long k, float re, float im, float& ore, float& oim) { // k = sample index + 1
        float cs, sn;
        rotor_at(ph0 + (unsigned long long)k * inc, cs, sn); // (ewise.hpp: ew_rotor is the same arithmetic)
        if (bad) cs = sn = __builtin_nanf("");
        ore = re * cs - im * sn;
        oim = re * sn + im * cs;
    };
    if constexpr (!V4) {
        for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
            const float2 v = reinterpret_cast<const float2*>(x)[i];
            float2       o;
            rot(i + 1, v.x, v.y, o.x, o.y);
            reinterpret_cast<float2*>(y)[i] = o;
        }
    } else
    for (long p = (long)blockIdx.x * blockDim.x + threadIdx.x; p < pairs; p += (long)gridDim.x * blockDim.x) {
        if (2 * p + 1 < n) {
            using f32x4 = __attribute__((ext_vector_type(4))) float;
            const f32x4 vv = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(x) + p);
            const float4 v = make_float4(vv[0], vv[1], vv[2], vv[3]);
            float4       o;
            rot(2 * p + 1, v.x, v.y, o.x, o.y);
            rot(2 * p + 2, v.z, v.w, o.z, o.w);
            const f32x4 oo = {o.x, o.y, o.z, o.w};
            __builtin_nontemporal_store(oo, reinterpret_cast<f32x4*>(y) + p);
        } else { // odd tail
            const float2 v = reinterpret_cast<const float2*>(x)[2 * p];
            float2       o;
            rot(2 * p + 1, v.x, v.y, o.x, o.y);
            reinterpret_cast<float2*>(y)[2 * p] = o;
        }
    }
}

// ------------------------------------------------------------------------------------------------ synthetic input
__device__ __forceinline__ uint64_t rotl64(uint64_t x, int k) { return (x << k) | (x >> (64 - k)); }
__device__ __forceinline__ uint64_t splitmix64(uint64_t& x) {
    uint64_t z = (x += 0x9e3779b97f4a7c15ULL);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ULL;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebULL;
    return z ^ (z >> 31);
}
struct Xo { uint64_t s0, s1, s2, s3; };
__device__ __forceinline__ uint64_t xo_next(Xo& s) { // Xoshiro256pp.hpp:41-51
    const uint64_t r = rotl64(s.s0 + s.s3, 23) + s.s0, t = s.s1 << 17;
    s.s2 ^= s.s0; s.s3 ^= s.s1; s.s1 ^= s.s2; s.s0 ^= s.s3; s.s2 ^= t; s.s3 = rotl64(s.s3, 45);
    return r;
}
__device__ __forceinline__ void polar_pair(Xo& s, float& g1, float& g2) { // GaussianNoise.hpp:101-111
    float u, v, q;
    do {
        u = 2.0f * ((float)(xo_next(s) >> 40) * 0x1.0p-24f) - 1.0f;
        v = 2.0f * ((float)(xo_next(s) >> 40) * 0x1.0p-24f) - 1.0f;
        q = u * u + v * v;
    } while (q >= 1.0f || q == 0.0f);
    const float f = sqrtf(-2.0f * logf(q) / q);
    g1 = u * f;
    g2 = v * f;
}

// the generator of 8-sample group g of a stream with seed `seed`: Xoshiro256pp(seed ^ kSynthMix (g + 1)) (constructor: Xoshiro256pp.hpp:33-39)
constexpr uint64_t kSynthMix = 0xd1b54a32d192ed03ULL;
__device__ __forceinline__ Xo synth_group_rng(uint64_t seed, uint64_t group) {
    uint64_t sm = seed ^ (kSynthMix * (group + 1));
    return Xo{splitmix64(sm), splitmix64(sm), splitmix64(sm), splitmix64(sm)};
}
__global__ void synth_draws_kernel(uint64_t* __restrict__ out, int n, uint64_t seed, uint64_t group) {
    if (threadIdx.x || blockIdx.x) return;
    Xo s = synth_group_rng(seed, group);
    for (int i = 0; i < n; ++i) out[i] = xo_next(s);
}

template <bool COMPLEX>
__global__ void synth_kernel(float* __restrict__ out, long n, uint64_t seed, double frel, float tone_amp, float noise_amp) {
    constexpr int PER    = 8; // samples per lane per visit
    const long    stride = (long)gridDim.x * blockDim.x * PER;
    for (long i0 = ((long)blockIdx.x * blockDim.x + threadIdx.x) * PER; i0 < n; i0 += stride) {
        Xo s = synth_group_rng(seed, (uint64_t)(i0 / PER));
        for (int k = 0; k < PER && i0 + k < n; COMPLEX ? ++k : k += 2) {
            float g1, g2;
            polar_pair(s, g1, g2);
            const long i = i0 + k;
            if constexpr (COMPLEX) {
                double sn, cs;
                const double ph = frel * (double)i;
                sincospi(2.0 * (ph - floor(ph)), &sn, &cs);
                out[2 * i]     = noise_amp * 0.70710678118654752440f * g1 + (float)(tone_amp * cs);
                out[2 * i + 1] = noise_amp * 0.70710678118654752440f * g2 + (float)(tone_amp * sn);
            } else {
                const double p0 = frel * (double)i, p1 = frel * (double)(i + 1);
                out[i] = noise_amp * g1 + (float)(tone_amp * sinpi(2.0 * (p0 - floor(p0))));
                if (i + 1 < n) out[i + 1] = noise_amp * g2 + (float)(tone_amp * sinpi(2.0 * (p1 - floor(p1))));
            }
        }
    }
}

} // namespace gr4

using namespace gr4;

struct gr4hip_rotator {
    float        inc  = 0.f;
    int          algo = GR4HIP_ROTATOR_CLOSED_FORM;
    int          cur  = 0; // state slot in use
    unsigned long long ph_fix = 0; // closed form: the carried phase as a 64-bit fraction of a turn, advanced on the host (inc is a host constant: nothing to read back,
                                   // and no rounding between calls -- one float per call is a 2.4e-7 rad random walk, beyond 1e-5 after a few thousand small calls)
    bool               ph_bad = false; // a non-finite initial phase: NaN out
    DeviceBuffer d_state; // recurrence: _accumulated_phase as the reference's float
    DeviceBuffer d_ckpt;
    float*       state() const { return static_cast<float*>(d_state.ptr) + cur; }
    // the stream rule (common.hpp): create / reset / set_algo note the phase the device word has to hold; the next call of the recurrence stores it on its own stream
    bool         store_pending = false;
    float        store_value   = 0.f;
};
namespace gr4 {
__global__ void rotator_store_phase_kernel(float* state, float value) { *state = value; }
}

extern "C" {

int gr4hip_math_const(int op, int dtype, const void* d_in, void* d_out, size_t n, const void* h_value, gr4hip_stream_t stream) {
    GR4_REQUIRE(h_value, "math_const: null value");
    if (n == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in && d_out, "math_const: null device pointer");
    NaryPtrs ins{};
    ins.p[0] = d_in;
    return math_dispatch<true>(op, dtype, ins, 1, h_value, d_out, (long)n, as_stream(stream));
}

int gr4hip_math_nary(int op, int dtype, const void* const* h_d_ins, size_t n_inputs, void* d_out, size_t n, gr4hip_stream_t stream) {
    GR4_REQUIRE(h_d_ins && n_inputs >= 1 && n_inputs <= (size_t)kMaxInputs, "math_nary: n_inputs must be in [1,32] (Math.hpp:90), got %zu", n_inputs);
    if (n == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_out, "math_nary: null output pointer");
    NaryPtrs ins{};
    for (size_t k = 0; k < n_inputs; ++k) {
        GR4_REQUIRE(h_d_ins[k], "math_nary: input %zu is a null device pointer", k);
        ins.p[k] = h_d_ins[k];
    }
    return math_dispatch<false>(op, dtype, ins, (int)n_inputs, nullptr, d_out, (long)n, as_stream(stream));
}

int gr4hip_rotator_create(gr4hip_rotator_t** out, float phase_increment, float initial_phase) {
    GR4_REQUIRE(out, "rotator: null output handle");
    auto* r = new (std::nothrow) gr4hip_rotator();
    GR4_REQUIRE(r, "out of host memory");
    r->inc = phase_increment;
    int rc = r->d_state.ensure(2 * sizeof(float));
    if (rc) { delete r; return rc; }
    rc = gr4hip_rotator_reset(r, initial_phase);
    if (rc) { delete r; return rc; }
    *out = r;
    return GR4HIP_OK;
}

int gr4hip_rotator_reset(gr4hip_rotator_t* r, float initial_phase) {
    GR4_REQUIRE(r, "rotator: null handle");
    r->store_pending = true; // (a launch of the recurrence still in flight on the caller's stream updates the word: the new value goes behind it, on the next call's stream)
    r->store_value   = initial_phase;
    r->ph_bad = !std::isfinite(initial_phase);
    r->ph_fix = r->ph_bad ? 0 : turns_fix((double)initial_phase / 6.283185307179586476925286766559);
    return GR4HIP_OK;
}

int gr4hip_rotator_set_algo(gr4hip_rotator_t* r, int algo) {
    GR4_REQUIRE(r, "rotator: null handle");
    GR4_REQUIRE(algo == GR4HIP_ROTATOR_CLOSED_FORM || algo == GR4HIP_ROTATOR_RECURRENCE, "rotator_set_algo: unknown algo %d", algo);
    if (algo == r->algo) return GR4HIP_OK;
    // the carried phase changes hands (rare: a settings change, not a per-call operation)
    if (algo == GR4HIP_ROTATOR_RECURRENCE) { // host -> device: stored in front of the next call, on its stream
        r->store_pending = true;
        r->store_value   = r->ph_bad ? std::nanf("") : (float)(fix_turns(r->ph_fix) * 6.283185307179586476925286766559);
    } else { // device -> host: everything queued on the handle, on whatever stream, finishes first
        float ph = r->store_value;
        if (!r->store_pending) {
            GR4_HIP_TRY(hipDeviceSynchronize());
            GR4_HIP_TRY(hipMemcpyAsync(&ph, r->state(), sizeof(float), hipMemcpyDeviceToHost, nullptr));
            GR4_HIP_TRY(hipStreamSynchronize(nullptr));
        }
        r->ph_bad = !std::isfinite(ph);
        r->ph_fix = r->ph_bad ? 0 : turns_fix((double)ph / 6.283185307179586476925286766559);
    }
    r->algo = algo;
    return GR4HIP_OK;
}

int gr4hip_rotator_process(gr4hip_rotator_t* r, const void* d_in, void* d_out, size_t n, gr4hip_stream_t stream) {
    GR4_REQUIRE(r, "rotator: null handle");
    if (n == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in && d_out, "rotator: null device pointer");
    hipStream_t  st      = as_stream(stream);
    if (r->algo == GR4HIP_ROTATOR_CLOSED_FORM && std::isfinite(r->inc)) { // (a NaN / infinite increment takes the recurrence: NaN out, like the reference)
        const unsigned long long inc = turns_fix((double)r->inc / 6.283185307179586476925286766559);
        const bool   v4    = (reinterpret_cast<uintptr_t>(d_in) & 15) == 0 && (reinterpret_cast<uintptr_t>(d_out) & 15) == 0;
        const size_t items = v4 ? (n + 1) / 2 : n;
        const unsigned grid = (unsigned)std::min<size_t>(ceil_div(items, (size_t)256), (size_t)1 << 20);
        if (v4) hipLaunchKernelGGL(rotator_closed_kernel<true>, dim3(grid), dim3(256), 0, st, (const float4*)d_in, (float4*)d_out, r->ph_fix, inc, (int)r->ph_bad, (long)n);
        else hipLaunchKernelGGL(rotator_closed_kernel<false>, dim3(grid), dim3(256), 0, st, (const float4*)d_in, (float4*)d_out, r->ph_fix, inc, (int)r->ph_bad, (long)n);
        GR4_LAUNCH_CHECK();
        r->ph_fix += (unsigned long long)n * inc; // the phase the next call starts from (modulo one turn, exactly)
        return GR4HIP_OK;
    }
    // the reference's float recurrence, bit for bit
    if (r->store_pending) {
        hipLaunchKernelGGL(rotator_store_phase_kernel, dim3(1), dim3(1), 0, st, r->state(), r->store_value);
        GR4_LAUNCH_CHECK();
        r->store_pending = false;
    }
    const size_t nchunks = ceil_div(n, (size_t)kRotChunk);
    int          rc      = r->d_ckpt.ensure(nchunks * sizeof(float));
    if (rc) return rc;
    // small increments spend many samples per binade: leap; large ones change binade or wrap every few samples and the plain walker's 27-cycle step wins
    const bool leap = dev_switch(kDevRotatorLeap) ? true : (fabsf(r->inc) < kRotLeapBelow && !dev_switch(kDevRotatorWalk)); // (developer / test switches)
    if (leap) hipLaunchKernelGGL(rotator_checkpoint_leap_kernel, dim3(1), dim3(64), 0, st, r->state(), r->inc, (float*)r->d_ckpt.ptr, (long)n);
    else if (r->inc >= 0.f) hipLaunchKernelGGL(rotator_checkpoint_kernel<1>, dim3(1), dim3(64), 0, st, r->state(), r->inc, (float*)r->d_ckpt.ptr, (long)n);
    else if (r->inc < 0.f) hipLaunchKernelGGL(rotator_checkpoint_kernel<-1>, dim3(1), dim3(64), 0, st, r->state(), r->inc, (float*)r->d_ckpt.ptr, (long)n);
    else hipLaunchKernelGGL(rotator_checkpoint_kernel<0>, dim3(1), dim3(64), 0, st, r->state(), r->inc, (float*)r->d_ckpt.ptr, (long)n); // NaN increment
    GR4_LAUNCH_CHECK();
    hipLaunchKernelGGL(rotator_apply_kernel, dim3((unsigned)ceil_div(nchunks, (size_t)256)), dim3(256), 0, st, (const float2*)d_in, (float2*)d_out,
                       (const float*)r->d_ckpt.ptr, r->inc, (long)n);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

int gr4hip_rotator_phase(gr4hip_rotator_t* r, float* phase, gr4hip_stream_t stream) {
    GR4_REQUIRE(r && phase, "rotator_phase: null argument");
    if (r->algo == GR4HIP_ROTATOR_CLOSED_FORM && std::isfinite(r->inc)) { // known on the host: the phase behind everything queued so far
        *phase = r->ph_bad ? std::nanf("") : (float)(fix_turns(r->ph_fix) * 6.283185307179586476925286766559);
        return GR4HIP_OK;
    }
    if (r->store_pending) { *phase = r->store_value; return GR4HIP_OK; } // (reset since the last call: the word on the device is not it yet)
    GR4_HIP_TRY(hipMemcpyAsync(phase, r->state(), sizeof(float), hipMemcpyDeviceToHost, as_stream(stream)));
    GR4_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return GR4HIP_OK;
}

int gr4hip_rotator_destroy(gr4hip_rotator_t* r) { delete r; return GR4HIP_OK; }

int gr4hip_synth_c32(void* d_out, size_t n, uint64_t seed, double tone_frel, float tone_amp, float noise_amp, gr4hip_stream_t stream) {
    if (n == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_out, "synth: null output");
    const unsigned grid = (unsigned)std::min<size_t>(ceil_div(n, (size_t)256 * 8), 8192);
    hipLaunchKernelGGL(synth_kernel<true>, dim3(grid), dim3(256), 0, as_stream(stream), (float*)d_out, (long)n, seed, tone_frel, tone_amp, noise_amp);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

int gr4hip_synth_draws(uint64_t* d_out, size_t n_draws, uint64_t seed, uint64_t group, gr4hip_stream_t stream) {
    if (n_draws == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_out && n_draws <= (1u << 20), "synth_draws: null output or more than 2^20 draws");
    hipLaunchKernelGGL(synth_draws_kernel, dim3(1), dim3(64), 0, as_stream(stream), d_out, (int)n_draws, seed, group);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

int gr4hip_synth_f32(float* d_out, size_t n, uint64_t seed, double tone_frel, float tone_amp, float noise_amp, gr4hip_stream_t stream) {
    if (n == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_out, "synth: null output");
    const unsigned grid = (unsigned)std::min<size_t>(ceil_div(n, (size_t)256 * 8), 8192);
    hipLaunchKernelGGL(synth_kernel<false>, dim3(grid), dim3(256), 0, as_stream(stream), d_out, (long)n, seed, tone_frel, tone_amp, noise_amp);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

} // extern "C"
