// common.hpp -- shared helpers of libgr4hip (gfx950 only; no CUDA/compat paths).
#pragma once
#include <cmath>
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <new>
#include <vector>

#include "../../include/gr4hip.h"

namespace gr4 {

void set_error(const char* fmt, ...);

// Developer switches (kernel A/B comparisons in the tests and tools; none of them changes what a call computes beyond rounding): read from the environment
// ONCE, when the library is loaded, and settable afterwards through gr4hip_developer_switch (atomic: no getenv on the call path, no race against setenv).
enum DevSwitch {
    kDevFirNoBf16x3 = 0,      // GR4HIP_FIR_NO_BF16X3: float32 multiply-add FIR kernels instead of the three-term bf16 ones (per handle: GR4HIP_FIR_EXACT_F32)
    kDevFirNoDecimFd,         // GR4HIP_FIR_NO_DECIM_FD: polyphase decimators instead of the frequency-domain decimate-by-8 kernel
    kDevIirThreePass,         // GR4HIP_IIR_THREE_PASS
    kDevIirLookback,          // GR4HIP_IIR_LOOKBACK
    kDevIirNoSplit,           // GR4HIP_IIR_NO_SPLIT
    kDevFftBluesteinPipeline, // GR4HIP_FFT_BLUESTEIN_PIPELINE
    kDevFftNoPipeline,        // GR4HIP_FFT_NO_PIPELINE
    kDevRotatorLeap,          // GR4HIP_ROTATOR_LEAP
    kDevRotatorWalk,          // GR4HIP_ROTATOR_WALK
    kDevChain16,              // GR4HIP_CHAIN16
    kDevFftSmoothRuntime,     // GR4HIP_FFT_SMOOTH_RUNTIME: the run-time mixed-radix kernel also for sizes that have a compile-time plan
    kDevEwiseNoDivRcp,        // GR4HIP_EWISE_NO_DIV_RCP: element-wise programs divide by a float constant with the general quotient (the tests compare it with the reciprocal form)
    kDevFirNoF16x2,           // GR4HIP_FIR_NO_F16X2: the three-term bf16 FIR kernels where the default takes the two-term f16 ones (per handle: GR4HIP_FIR_TIME_DOMAIN_BF16X3)
    kDevFirNoDecimF16,        // GR4HIP_FIR_NO_DECIM_F16: the frequency-domain decimate-by-8 kernel where the default takes the f16 band-form one (fir_decim_f16.hip)
    kDevFftBluesteinGeneric,  // GR4HIP_FFT_BLUESTEIN_GENERIC: the chirp convolution on the run-time radix-8 passes (one frame per workgroup) instead of the compile-time 16 x 16 x R3 plan
    kDevFftFourStep64k,       // GR4HIP_FFT_FOUR_STEP_64K: 65536-point transforms through the three-kernel four-step pipeline instead of the two-kernel 256 x 256 form
    kDevSwitchCount
};
int dev_switch(DevSwitch s);

#define GR4_HIP_TRY(expr)                                                                              \
    do {                                                                                               \
        hipError_t e_ = (expr);                                                                        \
        if (e_ != hipSuccess) {                                                                        \
            ::gr4::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
            return GR4HIP_RUNTIME_ERROR;                                                               \
        }                                                                                              \
    } while (0)

#define GR4_REQUIRE(cond, ...)                \
    do {                                      \
        if (!(cond)) {                        \
            ::gr4::set_error(__VA_ARGS__);    \
            return GR4HIP_INVALID_ARGUMENT;   \
        }                                     \
    } while (0)

#define GR4_LAUNCH_CHECK()                                                                      \
    do {                                                                                        \
        hipError_t e_ = hipGetLastError();                                                      \
        if (e_ != hipSuccess) {                                                                 \
            ::gr4::set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return GR4HIP_RUNTIME_ERROR;                                                        \
        }                                                                                       \
    } while (0)

// A HIP call whose failure the caller can do nothing about (frees in destructors): the runtime's last-error word is sticky since ROCm 7 -- an ignored failure
// stays there until somebody calls hipGetLastError, and that somebody is GR4_LAUNCH_CHECK behind an innocent launch (round 6: hipHostUnregister of a ring with a copy
// still in flight -> "kernel launch failed: unknown error" in the next gr4hip_iir_create, one run in two).  So: ignored means cleared.
inline void hip_quiet(hipError_t e) { if (e != hipSuccess) (void)hipGetLastError(); }

inline hipStream_t as_stream(gr4hip_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

inline size_t dtype_size(int dtype) {
    static const size_t s[14] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 8, 16, 8, 16};
    return (dtype >= 0 && dtype < 14) ? s[dtype] : 0;
}

inline bool is_pow2(size_t n) { return n && !(n & (n - 1)); }
inline int  ilog2(size_t n) { int l = 0; while ((size_t(1) << l) < n) ++l; return l; }
template <typename T> inline T ceil_div(T a, T b) { return (a + b - 1) / b; }

// A rotator's phase in TURNS as a 64-bit binary fraction (units of 2^-64 turn): sums and products modulo 2^64 ARE the reduction modulo one turn, so
// phase(k) = p0 + k inc is exact however long the stream and however it is associated -- p0 + k inc at once, or the phase of the sample before plus inc: every kernel
// that evaluates a rotator (math.hip, the element-wise programs, the decimators that carry one as their load program) lands on the same 64 bits.  The increment itself
// is rounded to 2^-64 turn once, whatever its sign (exact for |inc| >= 2^-11 turn; below that 2^-65 turn per sample: 3e-8 turn after 2^40 samples).
inline unsigned long long turns_fix_pos(double turns) { // turns >= 0
    const double f = turns - std::floor(turns); // [0, 1)
    return f >= 0.0 && f < 1.0 ? (unsigned long long)(f * 18446744073709551616.0) : 0ull; // (not finite: the callers mark those)
}
//  a negative argument is converted by magnitude and negated modulo 2^64: `turns - floor(turns)` = 1 - |turns| would round a small negative increment to
// 2^-53 turn (5e-17 per sample: 3.5e-5 rad after 1e11 samples at -1e-4 rad / sample) where the positive one of the same size keeps 2^-64
inline unsigned long long turns_fix(double turns) { return turns < 0.0 ? 0ull - turns_fix_pos(-turns) : turns_fix_pos(turns); }
inline double fix_turns(unsigned long long p) { return (double)p * (1.0 / 18446744073709551616.0); }
#ifdef __HIPCC__
// exp(j 2 pi phase): the top 32 bits as a signed fraction of a turn in [-0.5, 0.5) -> float (24 bits: 2^-26 turn), then the hardware sine / cosine, which take turns
// (max |error| 1.25e-7 on [-0.5, 0.5), tools/ubench/native_sincos_accuracy.hip)
__device__ __forceinline__ void rotor_at(unsigned long long phase, float& cs, float& sn) {
    const float tf = (float)(int)(unsigned)(phase >> 32) * 0x1p-32f;
    sn = __builtin_amdgcn_sinf(tf);
    cs = __builtin_amdgcn_cosf(tf);
}
#endif

// Per-device one-time set-up (function attributes are per device; a process may drive several GPUs through gr4hip_set_device).
// current(&first, &dev): CU count of the calling thread's current device (0: query failed).  The first time a table sees a device it returns
// the NEGATED count with *first = true: the caller does its one-time set-up for that device and records it with done(dev, count).
struct PerDevice {
    static constexpr int kMax = 64;
    int                  n_cu[kMax] = {};
    int                  current(bool* first, int* dev_out = nullptr) {
        int dev = 0;
        if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= kMax) { *first = true; return 0; }
        if (dev_out) *dev_out = dev;
        *first = n_cu[dev] == 0;
        if (*first) {
            int n = 0;
            if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) return 0;
            return -n; // not recorded yet: the caller records it with done() after its own set-up succeeded
        }
        return n_cu[dev];
    }
    void done(int dev, int n) { if (dev >= 0 && dev < kMax) n_cu[dev] = n; }
};

// small RAII device buffer used inside handles (grow-only)
struct DeviceBuffer {
    void*  ptr   = nullptr;
    size_t bytes = 0;
    int    ensure(size_t need) {
        if (need <= bytes) return GR4HIP_OK;
        if (ptr) hip_quiet(hipFree(ptr));
        ptr   = nullptr;
        bytes = 0;
        GR4_HIP_TRY(hipMalloc(&ptr, need));
        bytes = need;
        return GR4HIP_OK;
    }
    void release() { if (ptr) hip_quiet(hipFree(ptr)); ptr = nullptr; bytes = 0; }
    ~DeviceBuffer() { release(); }
};

// THE STREAM RULE (include/gr4hip.h "Lifecycle calls").  A handle's device state is written in exactly two ways:
//   (i)  upload_fresh(): a blocking copy into a buffer NO LAUNCH HAS SEEN YET -- the tables *_create builds, and tables built on first use inside a process call
//        (a new DeviceBuffer, or one whose ensure() has just replaced it: hipFree waits for the device).  Complete when it returns, so visible to work enqueued
//        afterwards on any stream.
//   (ii) work enqueued on the stream of a process call: hipMemsetAsync / hipMemcpyAsync / kernels on `st`.
// reset / set_taps / set_prologue / set_algo never touch the device: they note what the state has to become (zero_hist, taps_dirty, ...) and the next process
// call applies it on ITS stream, in front of its own launches and therefore behind everything that stream still has in flight for the handle.  That is the
// reference's contract -- reset() / settingsChanged() run on the block's worker between two work() calls (Block.hpp:606, 916-917, 1296; Scheduler.hpp:1938-1951) --
// without a device-wide wait, and it holds on hipStreamNonBlocking streams, which the NULL stream does not order against (round 5's race: a NULL-stream hipMemset
// of the carried history overtaken by the previous launch's carry).  A host-to-device hipMemcpyAsync from pageable memory returns when the source has been
// consumed (the runtime stages it), so host vectors may be locals.  tests/test_abi_host.py greps for bare hipMemset( / hipMemcpy( in this directory.
inline hipError_t upload_fresh(void* dst, const void* src, size_t bytes) { return hipMemcpy(dst, src, bytes, hipMemcpyHostToDevice); }

// host-side restatement of gr::algorithm::window::create<float> (algorithm/.../fourier/window.hpp:69-183)
int make_window(int type, float* w, size_t n, float beta);
int make_window64(int type, double* w, size_t n, double beta); // create<double>

} // namespace gr4
