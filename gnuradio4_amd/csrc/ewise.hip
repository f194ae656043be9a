// ewise.hip -- a run of per-sample blocks as ONE launch (ewise.hpp): the stand-alone kernel and the gr4hip_ewise_* entry points.
//
// Replaces what Merge<MultiplyConst, "out", Merge<DivideConst, "out", AddConst, "in">, "in"> compiles to upstream (BlockMerging.hpp:126-240: one processOne through
// all parts, values in registers) for chains known only at run time: every lane loads 16-byte vectors (non-temporal, one contiguous slab per workgroup like
// math_kernel), walks the op list with the values in registers, stores.  HBM traffic is 2 sizeof(T) bytes per sample whatever the number of ops; the bound
// moves to the vector ALU only for long chains of wide or divided types.
#include "ewise.hpp"

#include <cmath>
#include <cstdlib>
#include <limits>

namespace gr4 {

typedef unsigned int ew_u32x4 __attribute__((ext_vector_type(4)));
template <typename T> union EwVec16 { ew_u32x4 u; T e[16 / sizeof(T)]; };

// 16-byte vectors a lane holds across the op loop: ~16 values for 4-byte and wider types (the op fetch and branch amortise over them), fewer for narrow types
// (a u8 vector is 16 values already)
template <typename T> constexpr int ew_slab() { return sizeof(T) == 1 ? 1 : sizeof(T) == 2 ? 2 : 4; }

template <typename T>
__global__ __launch_bounds__(256) void ewise_kernel(const T* __restrict__ in, T* __restrict__ out, long n, long head, long nvec, EwiseHook prog) {
    // elements [head, head + nvec VE) are 16-byte aligned in both streams and take the vector body; the rest the scalar loop (a ring span may start at any element)
    constexpr int VE = 16 / sizeof(T), SL = ew_slab<T>(), NE = VE * SL;
    const long    v0 = (long)blockIdx.x * (256 * SL) + threadIdx.x;
    if (v0 < nvec) {
        EwVec16<T> a[SL];
#pragma unroll
        for (int s = 0; s < SL; ++s) {
            const long v = v0 + (long)s * 256;
            if (v < nvec) a[s].u = __builtin_nontemporal_load(&reinterpret_cast<const ew_u32x4*>(in + head)[v]);
            else a[s].u = ew_u32x4{0u, 0u, 0u, 0u};
        }
        T e[NE];
#pragma unroll
        for (int s = 0; s < SL; ++s)
#pragma unroll
            for (int j = 0; j < VE; ++j) e[s * VE + j] = a[s].e[j];
        ewise_apply<T, NE>(e, prog.ops, prog.n_ops, prog.has_div, [&](int j) { return prog.pos + head + (v0 + (long)(j / VE) * 256) * VE + (j % VE); });
#pragma unroll
        for (int s = 0; s < SL; ++s) {
#pragma unroll
            for (int j = 0; j < VE; ++j) a[s].e[j] = e[s * VE + j];
            const long v = v0 + (long)s * 256;
            if (v < nvec) __builtin_nontemporal_store(a[s].u, &reinterpret_cast<ew_u32x4*>(out + head)[v]);
        }
    }
    const long body_end = head + nvec * VE, nscalar = head + (n - body_end);
    for (long j = (long)blockIdx.x * blockDim.x + threadIdx.x; j < nscalar; j += (long)gridDim.x * blockDim.x) { // head and tail elements
        const long i = j < head ? j : body_end + (j - head);
        T          e[1] = {in[i]};
        ewise_apply<T, 1>(e, prog.ops, prog.n_ops, prog.has_div, [&](int) { return prog.pos + i; });
        out[i] = e[0];
    }
}

// Decimator<T> (time_domain_filter.hpp:234-244: keep the samples with i % decim == 0) with the per-sample blocks behind it in the same launch: only the kept samples
// are read at all.  (Memoryless blocks in FRONT of a Decimator commute with it -- f(x)[m D] == f(x[m D]) -- so the planner moves them behind and they land here too.)
template <typename T>
__global__ __launch_bounds__(256) void ewise_decimate_kernel(const T* __restrict__ in, T* __restrict__ out, long n_out, long decim, EwiseHook prog) {
    for (long m = (long)blockIdx.x * blockDim.x + threadIdx.x; m < n_out; m += (long)gridDim.x * blockDim.x) {
        T e[1] = {in[m * decim]};
        ewise_apply<T, 1>(e, prog.ops, prog.n_ops, prog.has_div, [&](int) { return prog.pos + m; });
        out[m] = e[0];
    }
}

template <typename T>
static int ewise_launch(const void* in, void* out, long n, const EwiseHook& prog, hipStream_t st) {
    constexpr long VE = 16 / sizeof(T), SL = ew_slab<T>();
    const auto mis  = [](const void* p) { return (long)((16 - (reinterpret_cast<uintptr_t>(p) & 15)) & 15); };
    long       head = mis(out);
    const bool same = head % (long)sizeof(T) == 0 && mis(in) == head;
    head            = same ? std::min<long>(head / (long)sizeof(T), n) : 0;
    const long nvec = same ? (n - head) / VE : 0;
    GR4_REQUIRE(ceil_div(nvec + 1, 256L * SL) < (1L << 31), "ewise: span too long for one launch");
    const long     nscalar = n - nvec * VE;
    const unsigned grid    = (unsigned)std::max<long>({ceil_div(nvec, 256L * SL), std::min<long>(ceil_div(nscalar, 1024L), 16384L), 1L});
    hipLaunchKernelGGL(ewise_kernel<T>, dim3(grid), dim3(256), 0, st, static_cast<const T*>(in), static_cast<T*>(out), n, head, nvec, prog);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

} // namespace gr4

using namespace gr4;

namespace gr4 {
template <typename F>
static bool exact_reciprocal(F c, F* r) { // c = +-2^k with a normal reciprocal: x / c and x * (1 / c) are the correctly rounded value of the same real number
    int     ex = 0;
    const F m  = std::frexp(c, &ex);
    if (std::fabs(m) != F(0.5) || ex <= std::numeric_limits<F>::min_exponent + 2 || ex >= std::numeric_limits<F>::max_exponent - 2) return false;
    *r = F(1) / c;
    return true;
}
template <typename F>
static void compile_float(gr4hip_ewise* p) {
    F    m = F(1), a = -F(0);
    bool have_m = false, have_a = false;
    auto flush = [&](const F* c) {
        if (!have_m && !have_a && !c) return;
        EwiseOp o{};
        o.kind = kEwAffine;
        std::memcpy(o.u.raw, &m, sizeof(F));
        std::memcpy(o.u.raw + 8, &a, sizeof(F));
        if (c) {
            o.flags = kEwFlagDiv;
            std::memcpy(o.u.raw + 16, c, sizeof(F));
            p->has_div = 1;
            if constexpr (std::is_same_v<F, float>) { // a constant divisor in a safe range: the device multiplies by its correctly rounded reciprocal and corrects (ewise.hpp)
                uint32_t bits;
                std::memcpy(&bits, c, 4);
                const float ac = std::fabs(*c);
                if (ac >= 0x1p-40f && ac <= 0x1p40f && (bits & 0x7fffffu) != 0x7fffffu && !dev_switch(kDevEwiseNoDivRcp)) {
                    const float y = 1.0f / *c; // IEEE division on the host: correctly rounded
                    o.flags |= kEwFlagDivRcp;
                    std::memcpy(o.u.raw + 20, &y, 4);
                }
            }
        }
        p->ops.push_back(o);
        m = F(1); a = -F(0); have_m = have_a = false;
    };
    for (const EwiseOp& u : p->user) {
        F c, r;
        std::memcpy(&c, u.u.raw, sizeof(F));
        int kind = u.kind;
        if (kind == kEwDiv && exact_reciprocal(c, &r)) { kind = kEwMul; c = r; }
        if (kind == kEwMul) { if (have_m || have_a) flush(nullptr); m = c; have_m = true; }
        else if (kind == kEwAdd || kind == kEwSub) { if (have_a) flush(nullptr); a = kind == kEwAdd ? c : -c; have_a = true; }
        else flush(&c);
    }
    flush(nullptr);
}
template <typename T>
static void compile_int(gr4hip_ewise* p) { // +, -, * modulo 2^w: one (M, A) per run, whatever its length
    uint64_t M = 1, A = 0;
    bool     any = false;
    auto     flush = [&](const T* c) {
        if (!any && !c) return;
        EwiseOp o{};
        o.kind = kEwAffine;
        const T m = (T)M, a = (T)A;
        std::memcpy(o.u.raw, &m, sizeof(T));
        std::memcpy(o.u.raw + 8, &a, sizeof(T));
        if (c) { o.flags = kEwFlagDiv; std::memcpy(o.u.raw + 16, c, sizeof(T)); p->has_div = 1; }
        p->ops.push_back(o);
        M = 1; A = 0; any = false;
    };
    for (const EwiseOp& u : p->user) {
        T c;
        std::memcpy(&c, u.u.raw, sizeof(T));
        const uint64_t v = (uint64_t)(int64_t)c; // sign-extended for signed T: the same residue modulo 2^w
        if (u.kind == kEwAdd) { A += v; any = true; }
        else if (u.kind == kEwSub) { A -= v; any = true; }
        else if (u.kind == kEwMul) { M *= v; A *= v; any = true; }
        else flush(&c);
    }
    flush(nullptr);
}
static void ewise_compile(gr4hip_ewise* p) {
    p->ops.clear();
    p->has_div = 0;
    switch (p->dtype) {
    case GR4HIP_U8: compile_int<uint8_t>(p); break;
    case GR4HIP_U16: compile_int<uint16_t>(p); break;
    case GR4HIP_U32: compile_int<uint32_t>(p); break;
    case GR4HIP_U64: compile_int<uint64_t>(p); break;
    case GR4HIP_I8: compile_int<int8_t>(p); break;
    case GR4HIP_I16: compile_int<int16_t>(p); break;
    case GR4HIP_I32: compile_int<int32_t>(p); break;
    case GR4HIP_I64: compile_int<int64_t>(p); break;
    case GR4HIP_F32: compile_float<float>(p); break;
    case GR4HIP_F64: compile_float<double>(p); break;
    default: p->ops = p->user; break; // complex: one item per op
    }
}
} // namespace gr4

namespace gr4 {
// (library-internal) the device copy of a program, uploaded on first use after a change
int ewise_device_ops(gr4hip_ewise* p, EwiseHook* hook, hipStream_t st) {
    if (p->dirty) ewise_compile(p);
    if (p->dirty && !p->ops.empty()) {
        int rc = p->d_ops.ensure(p->ops.size() * sizeof(EwiseOp));
        if (rc) return rc;
        GR4_HIP_TRY(hipMemcpyAsync(p->d_ops.ptr, p->ops.data(), p->ops.size() * sizeof(EwiseOp), hipMemcpyHostToDevice, st)); // (behind the launches `st` still has in flight with the old program)
    }
    p->dirty      = false;
    hook->ops     = p->ops.empty() ? nullptr : as_prog(p->d_ops.ptr);
    hook->n_ops   = (int)p->ops.size();
    hook->has_div = p->has_div;
    hook->pos     = p->pos;
    hook->rotor_only = p->dtype == GR4HIP_C32 && p->ops.size() == 1 && p->ops[0].kind == kEwRotate && !(p->ops[0].flags & kEwFlagNan);
    hook->rot_p0     = hook->rotor_only ? p->ops[0].u.q[0] : 0;
    hook->rot_inc    = hook->rotor_only ? p->ops[0].u.q[1] : 0;
    return GR4HIP_OK;
}
// (library-internal) a program that is nothing but real gains -- MultiplyConst / DivideConst on float, or on complex<float> with a real value: its product, in float64.
// A linear block absorbs such a neighbour into its coefficients (fir(g x) == (g b) * x).
bool ewise_as_real_gain(const gr4hip_ewise* p, double* gain) {
    double g = 1.0;
    for (const EwiseOp& op : p->user) {
        if (op.kind != kEwMul && op.kind != kEwDiv) return false;
        double v;
        if (p->dtype == GR4HIP_F32) { float f; std::memcpy(&f, op.u.raw, 4); v = f; }
        else if (p->dtype == GR4HIP_C32) { float f[2]; std::memcpy(f, op.u.raw, 8); if (f[1] != 0.f) return false; v = f[0]; }
        else return false;
        if (!(v == v) || std::isinf(v) || v == 0.0) return false;
        g = op.kind == kEwMul ? g * v : g / v;
    }
    *gain = g;
    return true;
}
// (library-internal) a program as its own launch on values of `dtype`: hooks of kernels that run it in front of / behind themselves
int ewise_run(const EwiseHook& prog, int dtype, const void* in, void* out, long n, hipStream_t st) {
    if (n <= 0 || prog.n_ops <= 0) return GR4HIP_OK;
    switch (dtype) {
    case GR4HIP_F32: return ewise_launch<float>(in, out, n, prog, st);
    case GR4HIP_C32: return ewise_launch<float2>(in, out, n, prog, st);
    case GR4HIP_F64: return ewise_launch<double>(in, out, n, prog, st);
    default: set_error("ewise_run: dtype %d", dtype); return GR4HIP_INVALID_ARGUMENT;
    }
}
gr4hip_ewise* ewise_clone(const gr4hip_ewise* p) { // a private copy of the op list (its own device buffer, position 0)
    auto* q = new (std::nothrow) gr4hip_ewise();
    if (q) { q->dtype = p->dtype; q->user = p->user; }
    return q;
}
} // namespace gr4

extern "C" {

int gr4hip_ewise_create(gr4hip_ewise_t** out, int dtype) {
    GR4_REQUIRE(out, "ewise: null output handle");
    GR4_REQUIRE(dtype_size(dtype), "ewise: unknown dtype %d", dtype);
    if (dtype == GR4HIP_UF32 || dtype == GR4HIP_UF64) { // (value + uncertainty pairs: the math entry points take them, one launch per block)
        set_error("ewise: programs of UncertainValue elements are not implemented");
        return GR4HIP_UNSUPPORTED;
    }
    auto* p = new (std::nothrow) gr4hip_ewise();
    GR4_REQUIRE(p, "out of host memory");
    p->dtype = dtype;
    *out     = p;
    return GR4HIP_OK;
}

int gr4hip_ewise_append_const(gr4hip_ewise_t* p, int op, const void* h_value) {
    GR4_REQUIRE(p && h_value, "ewise_append_const: null argument");
    GR4_REQUIRE(op >= GR4HIP_ADD && op <= GR4HIP_DIV, "ewise_append_const: unknown op %d", op);
    EwiseOp o{};
    o.kind = op;
    std::memcpy(o.u.raw, h_value, dtype_size(p->dtype));
    p->user.push_back(o);
    p->dirty = true;
    return GR4HIP_OK;
}

int gr4hip_ewise_append_rotator(gr4hip_ewise_t* p, float phase_increment, float initial_phase) {
    GR4_REQUIRE(p, "ewise_append_rotator: null handle");
    GR4_REQUIRE(p->dtype == GR4HIP_C32, "ewise_append_rotator: Rotator<complex<float>> needs a C32 program (dtype %d)", p->dtype);
    GR4_REQUIRE(phase_increment == phase_increment && initial_phase == initial_phase, "ewise_append_rotator: NaN phase (the stand-alone rotator's recurrence reproduces the reference there)");
    EwiseOp      o{};
    const double two_pi = 6.283185307179586476925286766559;
    const double ph_t = (double)initial_phase / two_pi, inc_t = (double)phase_increment / two_pi;
    o.kind = kEwRotate;
    if (std::isfinite(ph_t) && std::isfinite(inc_t)) {
        o.u.q[0] = turns_fix(ph_t);
        o.u.q[1] = turns_fix(inc_t);
    } else o.flags = kEwFlagNan;
    p->user.push_back(o);
    p->dirty = true;
    return GR4HIP_OK;
}

int gr4hip_ewise_length(const gr4hip_ewise_t* p, size_t* n_ops) {
    GR4_REQUIRE(p && n_ops, "ewise_length: null argument");
    *n_ops = p->user.size();
    return GR4HIP_OK;
}

int gr4hip_ewise_reset(gr4hip_ewise_t* p) {
    GR4_REQUIRE(p, "ewise_reset: null handle");
    p->pos = 0;
    return GR4HIP_OK;
}

int gr4hip_ewise_position(const gr4hip_ewise_t* p, uint64_t* samples) {
    GR4_REQUIRE(p && samples, "ewise_position: null argument");
    *samples = (uint64_t)p->pos;
    return GR4HIP_OK;
}

int gr4hip_ewise_process(gr4hip_ewise_t* p, const void* d_in, void* d_out, size_t n, gr4hip_stream_t stream) {
    GR4_REQUIRE(p, "ewise_process: null handle");
    if (n == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in && d_out, "ewise_process: null device pointer");
    EwiseHook prog;
    int       rc = ewise_device_ops(p, &prog, as_stream(stream));
    if (rc) return rc;
    hipStream_t st = as_stream(stream);
    if (prog.n_ops == 0) { // the empty program is the copy block
        if (d_in != d_out) GR4_HIP_TRY(hipMemcpyAsync(d_out, d_in, n * dtype_size(p->dtype), hipMemcpyDeviceToDevice, st));
        p->pos += (long)n;
        return GR4HIP_OK;
    }
    switch (p->dtype) {
    case GR4HIP_U8: rc = ewise_launch<uint8_t>(d_in, d_out, (long)n, prog, st); break;
    case GR4HIP_U16: rc = ewise_launch<uint16_t>(d_in, d_out, (long)n, prog, st); break;
    case GR4HIP_U32: rc = ewise_launch<uint32_t>(d_in, d_out, (long)n, prog, st); break;
    case GR4HIP_U64: rc = ewise_launch<uint64_t>(d_in, d_out, (long)n, prog, st); break;
    case GR4HIP_I8: rc = ewise_launch<int8_t>(d_in, d_out, (long)n, prog, st); break;
    case GR4HIP_I16: rc = ewise_launch<int16_t>(d_in, d_out, (long)n, prog, st); break;
    case GR4HIP_I32: rc = ewise_launch<int32_t>(d_in, d_out, (long)n, prog, st); break;
    case GR4HIP_I64: rc = ewise_launch<int64_t>(d_in, d_out, (long)n, prog, st); break;
    case GR4HIP_F32: rc = ewise_launch<float>(d_in, d_out, (long)n, prog, st); break;
    case GR4HIP_F64: rc = ewise_launch<double>(d_in, d_out, (long)n, prog, st); break;
    case GR4HIP_C32: rc = ewise_launch<float2>(d_in, d_out, (long)n, prog, st); break;
    default: rc = ewise_launch<double2>(d_in, d_out, (long)n, prog, st); break;
    }
    if (rc) return rc;
    p->pos += (long)n;
    return GR4HIP_OK;
}

int gr4hip_ewise_decimate(gr4hip_ewise_t* p, const void* d_in, size_t n_in, size_t decim, void* d_out, size_t* n_out_p, gr4hip_stream_t stream) {
    GR4_REQUIRE(p, "ewise_decimate: null handle");
    GR4_REQUIRE(decim >= 1, "ewise_decimate: decim must be >= 1");
    const size_t n_out = ceil_div(n_in, decim); // i % decim == 0 for i in [0, n_in)
    if (n_out_p) *n_out_p = n_out;
    if (n_out == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in && d_out, "ewise_decimate: null device pointer");
    EwiseHook prog;
    int       rc = ewise_device_ops(p, &prog, as_stream(stream));
    if (rc) return rc;
    hipStream_t    st   = as_stream(stream);
    const unsigned grid = (unsigned)std::min<size_t>(ceil_div(n_out, (size_t)256), 8192);
#define GR4_EWD(ID, T) case ID: hipLaunchKernelGGL(ewise_decimate_kernel<T>, dim3(grid), dim3(256), 0, st, static_cast<const T*>(d_in), static_cast<T*>(d_out), (long)n_out, (long)decim, prog); break
    switch (p->dtype) {
        GR4_EWD(GR4HIP_U8, uint8_t); GR4_EWD(GR4HIP_U16, uint16_t); GR4_EWD(GR4HIP_U32, uint32_t); GR4_EWD(GR4HIP_U64, uint64_t);
        GR4_EWD(GR4HIP_I8, int8_t); GR4_EWD(GR4HIP_I16, int16_t); GR4_EWD(GR4HIP_I32, int32_t); GR4_EWD(GR4HIP_I64, int64_t);
        GR4_EWD(GR4HIP_F32, float); GR4_EWD(GR4HIP_F64, double); GR4_EWD(GR4HIP_C32, float2);
    default: hipLaunchKernelGGL(ewise_decimate_kernel<double2>, dim3(grid), dim3(256), 0, st, static_cast<const double2*>(d_in), static_cast<double2*>(d_out), (long)n_out, (long)decim, prog); break;
    }
#undef GR4_EWD
    GR4_LAUNCH_CHECK();
    p->pos += (long)n_out;
    return GR4HIP_OK;
}

int gr4hip_ewise_destroy(gr4hip_ewise_t* p) { delete p; return GR4HIP_OK; }

} // extern "C"
