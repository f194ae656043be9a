// runtime.hip -- error state, device/memory/stream/event helpers and the VMM double-mapped ring of libgr4hip.
#include "common.hpp"

#include <sys/mman.h>
#include <sys/syscall.h>
#include <unistd.h>

#include <atomic>
#include <cstdlib>

#include <cmath>
#include <limits>

namespace gr4 {

static thread_local char g_err[512] = "";

static const char* const kDevNames[kDevSwitchCount] = {"GR4HIP_FIR_NO_BF16X3", "GR4HIP_FIR_NO_DECIM_FD", "GR4HIP_IIR_THREE_PASS", "GR4HIP_IIR_LOOKBACK", "GR4HIP_IIR_NO_SPLIT",
                                                       "GR4HIP_FFT_BLUESTEIN_PIPELINE", "GR4HIP_FFT_NO_PIPELINE", "GR4HIP_ROTATOR_LEAP", "GR4HIP_ROTATOR_WALK", "GR4HIP_CHAIN16", "GR4HIP_FFT_SMOOTH_RUNTIME", "GR4HIP_EWISE_NO_DIV_RCP", "GR4HIP_FIR_NO_F16X2", "GR4HIP_FIR_NO_DECIM_F16", "GR4HIP_FFT_BLUESTEIN_GENERIC", "GR4HIP_FFT_FOUR_STEP_64K"};
struct DevTable {
    std::atomic<int> v[kDevSwitchCount];
    DevTable() {
        for (int i = 0; i < kDevSwitchCount; ++i) {
            const char* e = std::getenv(kDevNames[i]);
            v[i].store(e ? (e[0] >= '0' && e[0] <= '9' ? std::atoi(e) : 1) : 0, std::memory_order_relaxed);
        }
    }
};
static DevTable& dev_table() { static DevTable t; return t; } // (initialised on first use: thread-safe, and independent of static-initialisation order)
int dev_switch(DevSwitch s) { return dev_table().v[s].load(std::memory_order_relaxed); }

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

template <typename T>
static T bessel_i0f(T x) { // window.hpp:42-56
    T       sum = 1, term = 1;
    int     k = 1;
    const T x_half = x / 2;
    do {
        term *= (x_half / static_cast<T>(k));
        sum += term * term;
        ++k;
    } while (term * term > sum * std::numeric_limits<T>::epsilon());
    return sum;
}

template <typename T>
static int make_window_t(int type, T* w, size_t n, T beta) { // gr::algorithm::window::create<T> (window.hpp:69-183), T = float | double
    if (n == 0) return GR4HIP_OK; // window.hpp:72-74
    const T pi2 = 2 * static_cast<T>(3.14159265358979323846);
    const T a   = pi2 / static_cast<T>(n - 1);
    auto    fill = [&](auto&& f) { for (size_t i = 0; i < n; ++i) w[i] = f(i); };
    switch (type) {
    case GR4HIP_WIN_NONE:
    case GR4HIP_WIN_RECTANGULAR: fill([](size_t) { return T(1); }); break;
    case GR4HIP_WIN_HAMMING: fill([a](size_t i) { return T(0.53836) - T(0.46164) * std::cos(a * T(i)); }); break;
    case GR4HIP_WIN_HANN: fill([a](size_t i) { return T(.5) - T(.5) * std::cos(a * T(i)); }); break;
    case GR4HIP_WIN_HANNEXP: fill([a](size_t i) { return std::pow(std::sin(a * T(i)), T(2.)); }); break;
    case GR4HIP_WIN_BLACKMAN: fill([a](size_t i) { const T ai = a * T(i); return T(0.42) - T(0.5) * std::cos(ai) + T(0.08) * std::cos(T(2.) * ai); }); break;
    case GR4HIP_WIN_NUTTALL: fill([a](size_t i) { const T ai = a * T(i); return T(0.355768) - T(0.487396) * std::cos(ai) + T(0.144232) * std::cos(2 * ai) - T(0.012604) * std::cos(3 * ai); }); break;
    case GR4HIP_WIN_BLACKMANHARRIS: fill([a](size_t i) { const T ai = a * T(i); return T(0.35875) - T(0.48829) * std::cos(ai) + T(0.14128) * std::cos(2 * ai) - T(0.01168) * std::cos(3 * ai); }); break;
    case GR4HIP_WIN_BLACKMANNUTTALL: fill([a](size_t i) { const T ai = a * T(i); return T(0.3635819) - T(0.4891775) * std::cos(ai) + T(0.1365995) * std::cos(T(2.) * ai) - T(0.0106411) * std::cos(T(3.) * ai); }); break;
    case GR4HIP_WIN_FLATTOP: fill([a](size_t i) { const T ai = a * T(i); return T(1.0) - T(1.93) * std::cos(ai) + T(1.29) * std::cos(2 * ai) - T(0.388) * std::cos(3 * ai) + T(0.032) * std::cos(4 * ai); }); break;
    case GR4HIP_WIN_EXPONENTIAL: { const T exp0 = std::exp(T(0.)); const T aa = T(3.) * T(n); fill([=](size_t i) { return std::exp(T(i) / aa) / exp0; }); break; }
    case GR4HIP_WIN_KAISER: {
        if (beta < 0 || n <= 1) { set_error("Kaiser window: beta must be >= 0 and n > 1"); return GR4HIP_INVALID_ARGUMENT; } // window.hpp:167-172 throws
        const T factor = T(1) / T(n - 1), i0Beta = bessel_i0f(beta);
        fill([=](size_t i) { const T term = (T(2 * i) * factor) - T(1); return bessel_i0f(beta * std::sqrt(std::abs(T(1) - term * term))) / i0Beta; });
        break;
    }
    default: set_error("unknown window type %d", type); return GR4HIP_INVALID_ARGUMENT;
    }
    return GR4HIP_OK;
}
int make_window(int type, float* w, size_t n, float beta) { return make_window_t<float>(type, w, n, beta); }
int make_window64(int type, double* w, size_t n, double beta) { return make_window_t<double>(type, w, n, beta); }

} // namespace gr4

using namespace gr4;

struct gr4hip_ring {
    void*                        base = nullptr;
    size_t                       size = 0;
    hipMemGenericAllocationHandle_t handle{};
    bool                         mapped[2] = {false, false};
    bool                         have_handle = false;
};

extern "C" {

int gr4hip_abi_version(void) { return GR4HIP_ABI_VERSION; }
const char* gr4hip_last_error(void) { return g_err; }

int gr4hip_developer_switch(const char* name, int value) {
    GR4_REQUIRE(name, "developer_switch: null name");
    for (int i = 0; i < gr4::kDevSwitchCount; ++i)
        if (std::strcmp(name, gr4::kDevNames[i]) == 0) { gr4::dev_table().v[i].store(value, std::memory_order_relaxed); return GR4HIP_OK; }
    gr4::set_error("developer_switch: unknown switch %s", name);
    return GR4HIP_INVALID_ARGUMENT;
}

const char* gr4hip_status_string(int s) {
    switch (s) {
    case GR4HIP_OK: return "OK";
    case GR4HIP_DONE: return "DONE";
    case GR4HIP_INSUFFICIENT_INPUT: return "INSUFFICIENT_INPUT_ITEMS";
    case GR4HIP_INSUFFICIENT_OUTPUT: return "INSUFFICIENT_OUTPUT_ITEMS";
    case GR4HIP_ERROR: return "ERROR";
    case GR4HIP_INVALID_ARGUMENT: return "INVALID_ARGUMENT";
    case GR4HIP_RUNTIME_ERROR: return "HIP_RUNTIME_ERROR";
    case GR4HIP_UNSUPPORTED: return "UNSUPPORTED";
    case GR4HIP_NO_DEVICE: return "NO_DEVICE";
    default: return "UNKNOWN";
    }
}

int gr4hip_device_count(int* count) {
    GR4_REQUIRE(count, "count is null");
    int        n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) { (void)hipGetLastError(); n = 0; }
    *count = n;
    return GR4HIP_OK;
}
int gr4hip_set_device(int index) {
    int n = 0;
    gr4hip_device_count(&n);
    if (n == 0) { set_error("no HIP device visible"); return GR4HIP_NO_DEVICE; }
    GR4_REQUIRE(index >= 0 && index < n, "device index %d out of range [0,%d)", index, n);
    GR4_HIP_TRY(hipSetDevice(index));
    return GR4HIP_OK;
}
int gr4hip_get_device(int* index) { GR4_REQUIRE(index, "index is null"); GR4_HIP_TRY(hipGetDevice(index)); return GR4HIP_OK; }
int gr4hip_device_name(int index, char* buf, size_t buflen) {
    GR4_REQUIRE(buf && buflen, "buf is null");
    hipDeviceProp_t p;
    GR4_HIP_TRY(hipGetDeviceProperties(&p, index));
    snprintf(buf, buflen, "%s (%s, %d CUs)", p.name, p.gcnArchName, p.multiProcessorCount);
    return GR4HIP_OK;
}

int gr4hip_malloc(void** d_ptr, size_t bytes) { GR4_REQUIRE(d_ptr, "d_ptr is null"); GR4_HIP_TRY(hipMalloc(d_ptr, bytes ? bytes : 1)); return GR4HIP_OK; }
int gr4hip_free(void* d_ptr) { if (d_ptr) GR4_HIP_TRY(hipFree(d_ptr)); return GR4HIP_OK; }
int gr4hip_malloc_host(void** h_ptr, size_t bytes) { GR4_REQUIRE(h_ptr, "h_ptr is null"); GR4_HIP_TRY(hipHostMalloc(h_ptr, bytes ? bytes : 1, hipHostMallocDefault)); return GR4HIP_OK; }
int gr4hip_free_host(void* h_ptr) { if (h_ptr) GR4_HIP_TRY(hipHostFree(h_ptr)); return GR4HIP_OK; }
int gr4hip_host_ring_create(void** base_out, size_t bytes) {
    GR4_REQUIRE(base_out, "host_ring_create: null output");
    const long page = sysconf(_SC_PAGESIZE);
    GR4_REQUIRE(bytes > 0 && page > 0 && bytes % (size_t)page == 0, "host_ring_create: %zu bytes is not a multiple of the page size", bytes);
    const int fd = (int)syscall(SYS_memfd_create, "gr4hip_host_ring", 1u /*MFD_CLOEXEC*/);
    if (fd < 0 || ftruncate(fd, (off_t)bytes) != 0) { if (fd >= 0) close(fd); set_error("host_ring_create: memfd of %zu bytes failed", bytes); return GR4HIP_RUNTIME_ERROR; }
    char* base = static_cast<char*>(mmap(nullptr, 2 * bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0)); // the address range, then the same pages into both halves
    bool  ok   = base != MAP_FAILED;
    ok = ok && mmap(base, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0) != MAP_FAILED;
    ok = ok && mmap(base + bytes, bytes, PROT_READ | PROT_WRITE, MAP_SHARED | MAP_FIXED, fd, 0) != MAP_FAILED;
    close(fd);
    if (!ok) { if (base != MAP_FAILED) munmap(base, 2 * bytes); set_error("host_ring_create: double mapping of %zu bytes failed", bytes); return GR4HIP_RUNTIME_ERROR; }
    const hipError_t e = hipHostRegister(base, 2 * bytes, hipHostRegisterDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); munmap(base, 2 * bytes); set_error("host_ring_create: hipHostRegister failed: %s", hipGetErrorString(e)); return GR4HIP_RUNTIME_ERROR; }
    *base_out = base;
    return GR4HIP_OK;
}
int gr4hip_host_ring_destroy(void* base, size_t bytes) {
    if (!base) return GR4HIP_OK;
    // a copy engine may still be reading or writing these pages (the last chunks of an edge): unregistering under it fails ("unknown error", observed one run in two of
    // the host engine's test) and leaves the pages pinned.  Everything queued on the device finishes first -- this is a teardown call.
    hip_quiet(hipDeviceSynchronize());
    const hipError_t e = hipHostUnregister(base);
    hip_quiet(e);
    munmap(base, 2 * bytes);
    if (e != hipSuccess) { set_error("host_ring_destroy: hipHostUnregister failed: %s", hipGetErrorString(e)); return GR4HIP_RUNTIME_ERROR; }
    return GR4HIP_OK;
}
int gr4hip_memcpy_h2d(void* d, const void* h, size_t bytes, gr4hip_stream_t s) { if (bytes) GR4_HIP_TRY(hipMemcpyAsync(d, h, bytes, hipMemcpyHostToDevice, as_stream(s))); return GR4HIP_OK; }
int gr4hip_memcpy_d2h(void* h, const void* d, size_t bytes, gr4hip_stream_t s) { if (bytes) GR4_HIP_TRY(hipMemcpyAsync(h, d, bytes, hipMemcpyDeviceToHost, as_stream(s))); return GR4HIP_OK; }
int gr4hip_memcpy_d2d(void* dd, const void* ds, size_t bytes, gr4hip_stream_t s) { if (bytes) GR4_HIP_TRY(hipMemcpyAsync(dd, ds, bytes, hipMemcpyDeviceToDevice, as_stream(s))); return GR4HIP_OK; }
int gr4hip_memset(void* d, int v, size_t bytes, gr4hip_stream_t s) { if (bytes) GR4_HIP_TRY(hipMemsetAsync(d, v, bytes, as_stream(s))); return GR4HIP_OK; }
int gr4hip_stream_create(gr4hip_stream_t* s) { GR4_REQUIRE(s, "stream is null"); hipStream_t st; GR4_HIP_TRY(hipStreamCreateWithFlags(&st, hipStreamNonBlocking)); *s = st; return GR4HIP_OK; }
int gr4hip_stream_destroy(gr4hip_stream_t s) { if (s) GR4_HIP_TRY(hipStreamDestroy(as_stream(s))); return GR4HIP_OK; }
int gr4hip_stream_synchronize(gr4hip_stream_t s) { GR4_HIP_TRY(hipStreamSynchronize(as_stream(s))); return GR4HIP_OK; }
int gr4hip_stream_query(gr4hip_stream_t s, int* idle) {
    GR4_REQUIRE(idle, "idle is null");
    hipError_t e = hipStreamQuery(as_stream(s));
    if (e == hipSuccess) { *idle = 1; return GR4HIP_OK; }
    if (e == hipErrorNotReady) { (void)hipGetLastError(); *idle = 0; return GR4HIP_OK; }
    set_error("hipStreamQuery failed: %s", hipGetErrorString(e));
    return GR4HIP_RUNTIME_ERROR;
}
int gr4hip_event_create(gr4hip_event_t* ev) { GR4_REQUIRE(ev, "ev is null"); hipEvent_t e; GR4_HIP_TRY(hipEventCreate(&e)); *ev = e; return GR4HIP_OK; }
int gr4hip_event_destroy(gr4hip_event_t ev) { if (ev) GR4_HIP_TRY(hipEventDestroy((hipEvent_t)ev)); return GR4HIP_OK; }
int gr4hip_event_record(gr4hip_event_t ev, gr4hip_stream_t s) { GR4_HIP_TRY(hipEventRecord((hipEvent_t)ev, as_stream(s))); return GR4HIP_OK; }
int gr4hip_stream_wait_event(gr4hip_stream_t s, gr4hip_event_t ev) { GR4_HIP_TRY(hipStreamWaitEvent(as_stream(s), (hipEvent_t)ev, 0)); return GR4HIP_OK; }
int gr4hip_event_synchronize(gr4hip_event_t ev) { GR4_HIP_TRY(hipEventSynchronize((hipEvent_t)ev)); return GR4HIP_OK; }
int gr4hip_event_query(gr4hip_event_t ev, int* done) {
    GR4_REQUIRE(done, "done is null");
    hipError_t e = hipEventQuery((hipEvent_t)ev);
    if (e == hipSuccess) { *done = 1; return GR4HIP_OK; }
    if (e == hipErrorNotReady) { (void)hipGetLastError(); *done = 0; return GR4HIP_OK; }
    set_error("hipEventQuery failed: %s", hipGetErrorString(e));
    return GR4HIP_RUNTIME_ERROR;
}
int gr4hip_event_elapsed_ms(gr4hip_event_t a, gr4hip_event_t b, float* ms) { GR4_REQUIRE(ms, "ms is null"); GR4_HIP_TRY(hipEventElapsedTime(ms, (hipEvent_t)a, (hipEvent_t)b)); return GR4HIP_OK; }

// ---------------------------------------------------------------------------------------------- double-mapped ring
int gr4hip_ring_create(gr4hip_ring_t** out, size_t min_bytes) {
    GR4_REQUIRE(out && min_bytes, "ring: null output or zero size");
    int dev = 0;
    GR4_HIP_TRY(hipGetDevice(&dev));
    hipMemAllocationProp prop{};
    prop.type          = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id   = dev;
    size_t gran        = 0;
    GR4_HIP_TRY(hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended));
    if (gran == 0) gran = size_t(2) << 20;
    const size_t size = ceil_div(min_bytes, gran) * gran; // CircularBuffer.hpp:299-321 rounds to a page multiple likewise
    auto*        r    = new (std::nothrow) gr4hip_ring();
    GR4_REQUIRE(r, "out of host memory");
    r->size = size;
    auto fail = [&](const char* what, hipError_t e) {
        set_error("ring: %s failed: %s", what, hipGetErrorString(e));
        gr4hip_ring_destroy(r);
        return GR4HIP_RUNTIME_ERROR;
    };
    hipError_t e = hipMemAddressReserve(&r->base, 2 * size, gran, nullptr, 0);
    if (e != hipSuccess) { r->base = nullptr; return fail("hipMemAddressReserve", e); }
    e = hipMemCreate(&r->handle, size, &prop, 0);
    if (e != hipSuccess) return fail("hipMemCreate", e);
    r->have_handle = true;
    hipMemAccessDesc acc{};
    acc.location = prop.location;
    acc.flags    = hipMemAccessFlagsProtReadWrite;
    for (int k = 0; k < 2; ++k) {
        void* va = static_cast<char*>(r->base) + k * size;
        e        = hipMemMap(va, size, 0, r->handle, 0);
        if (e != hipSuccess) return fail("hipMemMap", e);
        r->mapped[k] = true;
        e            = hipMemSetAccess(va, size, &acc, 1);
        if (e != hipSuccess) return fail("hipMemSetAccess", e);
    }
    *out = r;
    return GR4HIP_OK;
}

int gr4hip_ring_destroy(gr4hip_ring_t* r) {
    if (!r) return GR4HIP_OK;
    for (int k = 0; k < 2; ++k)
        if (r->mapped[k]) hip_quiet(hipMemUnmap(static_cast<char*>(r->base) + k * r->size, r->size));
    if (r->have_handle) hip_quiet(hipMemRelease(r->handle));
    if (r->base) hip_quiet(hipMemAddressFree(r->base, 2 * r->size));
    delete r;
    return GR4HIP_OK;
}
int gr4hip_ring_base(const gr4hip_ring_t* r, void** d_base) { GR4_REQUIRE(r && d_base, "ring: null"); *d_base = r->base; return GR4HIP_OK; }
int gr4hip_ring_size(const gr4hip_ring_t* r, size_t* bytes) { GR4_REQUIRE(r && bytes, "ring: null"); *bytes = r->size; return GR4HIP_OK; }

int gr4hip_window_create(int window, float* h_out, size_t n, float beta) {
    GR4_REQUIRE(h_out || n == 0, "window: null output");
    return make_window(window, h_out, n, beta);
}
int gr4hip_window_create_f64(int window, double* h_out, size_t n, double beta) {
    GR4_REQUIRE(h_out || n == 0, "window: null output");
    return make_window64(window, h_out, n, beta);
}

} // extern "C"
