// gr4/hip.hpp -- the live compute_domain = "gpu:hip[:i]" seam: device implementations of the hot-path blocks on top of the C-ABI
// (include/gr4hip.h, libgr4hip.so) and the HIP-stream scheduler that fuses adjacent device blocks.
//
//   * gr::hip::Kernel<Block>::work  -- what Block::dispatchProcessing calls at the seam (core/include/gnuradio-4.0/Block.hpp:1855-1862)
//     when a single block is on the device inside a host graph: span -> pinned staging -> HBM -> kernel -> HBM -> span.
//   * gr::hip::StreamScheduler      -- scheduler::Simple plus device runs: a maximal linear chain of device blocks becomes ONE
//     work unit whose intermediate edges never leave HBM; samples arrive through pinned hipMemcpyAsync into a double-mapped
//     device ring (gr4hip_ring_*) and adjacent blocks are fused into one launch where a fused kernel exists
//     (fir_filter<complex<float>> -> PowerSpectrum == gr4hip_chain_*: the runtime analogue of Merge<>, BlockMerging.hpp:136-320).
// Device blocks never fall back to the host path: a failing library call turns into work::Status::ERROR with the library's text.
#pragma once
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdlib>
#include <cctype>
#include <cmath>
#include <cstring>
#include <deque>
#include <mutex>
#include <numeric>
#include <span>
#include <thread>
#include <functional>
#include <iostream>
#include <limits>
#include <memory_resource>
#include <new>

#include "../../../../include/gr4hip.h"
#include "blocks.hpp"
#include "merge.hpp"

namespace gr::hip {

inline void check(int rc, const char* what) {
    if (rc < 0 && rc != GR4HIP_DONE && rc != GR4HIP_INSUFFICIENT_INPUT && rc != GR4HIP_INSUFFICIENT_OUTPUT)
        throw std::runtime_error(std::string(what) + ": " + gr4hip_status_string(rc) + " (" + gr4hip_last_error() + ")");
}

// Process-wide choices of the device implementations that a block's reflected settings do not carry (the block definitions are the reference's: no extra
// members).  Set before the graph is initialised.
struct Options {
    // Rotator<complex<float>> on a device: false (default) = closed-form float64 phase -- the float64 oracle's output, one HBM-bound pass; it does NOT reproduce
    // the drift of the reference's float accumulator (beyond 1e-5 of the CPU block's own output after ~10^3 .. 10^5 samples).  true = the reference's float
    // recurrence itself, bit-identical to the block on the host (a sequential walk: far slower; include/gr4hip.h, gr4hip_rotator_set_algo).
    bool rotator_reference_recurrence = false;
    // Dynamic-range guard of the frequency-domain kernels (fused chain, long complex FIR spans, decimate-by-8 FIR): GR4HIP_GUARD_STRICT (default: the frames /
    // segments a launch marks are evaluated again in the time domain by launches enqueued behind it on the same stream -- nobody waits, since round 5),
    // GR4HIP_GUARD_DEFERRED (enqueues stay asynchronous, the switch lags by one chunk) or GR4HIP_GUARD_OFF (include/gr4hip.h)
    int guard_mode = GR4HIP_GUARD_STRICT;
    // A sharded graph's exchange (FanInRun) waits for other ranks: seconds one exchange may take before the run gives up with a rank-tagged message and
    // work::Status::ERROR instead of hanging its process (a peer that died or never joined); <= 0: wait for ever
    double collective_timeout_s = 120.0;
    // A device run whose input edge is smaller than a launch is worth (the reference's default: 65 536 items, Graph.hpp:102) gathers the edge's contents in its device
    // ring and launches over ~2^20 items at a time (DeviceRun "batching"); false: one launch per edge-full, round 5's behaviour (0.8 Gsamples/s host-fed at 2^16-item edges)
    bool batch_small_edges = true;
};
inline Options& options() { static Options o; return o; }

// A run of per-sample blocks -- MathOpImpl<T, op> (Math.hpp:38-56), Rotator<complex<float>> (Rotator.hpp:51-61) -- as data: what Merge<A, "out", B, "in">
// (BlockMerging.hpp:126-240) is at compile time upstream.  One program = one launch (gr4hip_ewise_*), or no launch at all when a neighbouring filter takes it
// as its load / store hook (Stage::absorb).
struct EwiseProgram {
    struct Op {
        int                           kind = GR4HIP_ADD; // gr4hip_op, or kRotator
        std::array<unsigned char, 16> value{};           // the constant, one element of the program's dtype
        float                         inc = 0.f, ph0 = 0.f; // kRotator
    };
    static constexpr int kRotator = 4;
    int                  dtype = GR4HIP_F32;
    std::vector<Op>      ops;
    [[nodiscard]] bool empty() const { return ops.empty(); }
    void append(const EwiseProgram& o) { ops.insert(ops.end(), o.ops.begin(), o.ops.end()); }
    void prepend(const EwiseProgram& o) { ops.insert(ops.begin(), o.ops.begin(), o.ops.end()); }
    [[nodiscard]] std::string describe() const {
        static constexpr const char* names[] = {"add", "sub", "mul", "div", "rot"};
        std::string d;
        if (ops.size() > 6) return std::to_string(ops.size()) + " ops";
        for (const auto& o : ops) d += std::string(d.empty() ? "" : ",") + names[std::clamp(o.kind, 0, 4)];
        return d;
    }
    // a fresh library handle (the caller destroys it; a filter that takes the program as a hook copies it)
    [[nodiscard]] gr4hip_ewise_t* make() const {
        gr4hip_ewise_t* h = nullptr;
        if (const int rc = gr4hip_ewise_create(&h, dtype); rc < 0) throw std::runtime_error(std::string("gr4hip_ewise_create: ") + gr4hip_last_error());
        for (const auto& o : ops) {
            const int rc = o.kind == kRotator ? gr4hip_ewise_append_rotator(h, o.inc, o.ph0) : gr4hip_ewise_append_const(h, o.kind, o.value.data());
            if (rc < 0) { gr4hip_ewise_destroy(h); throw std::runtime_error(std::string("gr4hip_ewise_append: ") + gr4hip_last_error()); }
        }
        return h;
    }
};

// the hooks of a stage that runs its neighbours' per-sample ops inside its own launch
struct AbsorbedPrograms {
    EwiseProgram pre, post;
    bool take(const EwiseProgram& p, bool before, int dtype) {
        if (p.dtype != dtype) return false;
        pre.dtype = post.dtype = dtype;
        if (before) pre.prepend(p); else post.append(p);
        return true;
    }
    void clear() { pre.ops.clear(); post.ops.clear(); }
    [[nodiscard]] std::string suffix() const {
        std::string s;
        if (!pre.empty()) s += "pre: " + pre.describe();
        if (!post.empty()) s += std::string(s.empty() ? "" : "; ") + "post: " + post.describe();
        return s.empty() ? s : "[" + s + "]";
    }
    // the product of a program that is nothing but real gains (what a linear stage folds into its coefficients); false otherwise
    static bool real_gain(const EwiseProgram& p, double* g) {
        double acc = 1.0;
        for (const auto& o : p.ops) {
            if (o.kind != GR4HIP_MUL && o.kind != GR4HIP_DIV) return false;
            float v[2] = {0.f, 0.f};
            if (p.dtype == GR4HIP_F32) std::memcpy(v, o.value.data(), 4);
            else if (p.dtype == GR4HIP_C32) std::memcpy(v, o.value.data(), 8);
            else return false;
            if (v[1] != 0.f || v[0] == 0.f || !std::isfinite(v[0])) return false;
            acc = o.kind == GR4HIP_MUL ? acc * double(v[0]) : acc / double(v[0]);
        }
        *g = acc;
        return true;
    }
};

// one device stage of a chain: consumes n_in elements at d_in, produces *n_out at d_out, asynchronously on `stream`
struct Stage {
    virtual ~Stage() = default;
    virtual int              enqueue(const void* d_in, std::size_t n_in, void* d_out, std::size_t* n_out, gr4hip_stream_t stream) = 0;
    virtual std::string_view kind() const                                                                                         = 0;
    std::size_t              in_bytes = 4, out_bytes = 4; // element sizes
    std::size_t              in_chunk = 1, out_chunk = 1; // whole chunks only (Resampling)
    // ---- kernel-level fusion (the run-time Merge<>): a stage that is nothing but per-sample ops says so; a stage that can run a neighbour's ops inside its own
    // launch takes them.  before: the program works on this stage's INPUT (in front of whatever it already does there), otherwise on its output (behind).
    [[nodiscard]] virtual const EwiseProgram* program() const { return nullptr; }
    virtual bool absorb(const EwiseProgram& /*p*/, bool /*before*/) { return false; }
    virtual void clear_absorbed() {}
    virtual Stage* self() { return this; } // (a wrapper that only owns a stage answers with the stage: fusion looks at concrete types)
};

// grow-only device / pinned buffers
struct DevBuf {
    void*       p = nullptr;
    std::size_t n = 0;
    bool        pinned;
    explicit DevBuf(bool pinned_ = false) : pinned(pinned_) {}
    DevBuf(const DevBuf&)            = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    void* ensure(std::size_t bytes) {
        if (bytes > n) {
            release();
            check(pinned ? gr4hip_malloc_host(&p, bytes) : gr4hip_malloc(&p, bytes), "device allocation");
            n = bytes;
        }
        return p;
    }
    void release() {
        if (p) (pinned ? gr4hip_free_host(p) : gr4hip_free(p));
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

// ---------------------------------------------------------------------------------------------- the "hip" memory provider (ComputeDomain.hpp:105-173)
// Edge storage for EdgeParameters{.domain = "gpu:hip[:i]"} on CPU-domain ports: page-locked host memory.  The samples stay addressable by the
// host blocks on both ends of the edge, and the copy engines read them in place -- gr::hip::H2D skips its staging copy for such an edge.
// (Edges between GPU-domain ports do not go through a memory_resource at all: DeviceEdgeBuffer below owns a VMM double mapping in HBM.)
class PinnedResource final : public std::pmr::memory_resource, public RingResource {
public:
    // (round 5) edges of this provider are page-locked RINGS where their size allows (gr4hip_host_ring_create: memfd + two mappings + hipHostRegister): a span that wraps
    // the end is contiguous, the copy engine takes it in place, nothing is ever moved to the front of the edge
    [[nodiscard]] void* ring_allocate(std::size_t bytes) override {
        void* p = nullptr;
        return (bytes % 4096 == 0 && gr4hip_host_ring_create(&p, bytes) == GR4HIP_OK) ? p : nullptr;
    }
    void ring_deallocate(void* base, std::size_t bytes) override { gr4hip_host_ring_destroy(base, bytes); }

private:
    void* do_allocate(std::size_t bytes, std::size_t align) override {
        void* p = nullptr;
        if (align > 4096 || gr4hip_malloc_host(&p, std::max<std::size_t>(bytes, 1)) != GR4HIP_OK) throw std::bad_alloc();
        return p;
    }
    void do_deallocate(void* p, std::size_t, std::size_t) override { gr4hip_free_host(p); }
    bool do_is_equal(const std::pmr::memory_resource& o) const noexcept override { return this == &o; }
};
inline std::pmr::memory_resource* pinned_resource() {
    static PinnedResource r;
    return &r;
}
inline void register_provider() { // idempotent; call once before Graph::connect with a "gpu:hip" edge domain
    ComputeRegistry::instance().register_provider("hip", [](const ComputeDomain& dom, void*) -> std::pmr::memory_resource* {
        if (dom.kind != "gpu") return nullptr;
        if (gr4hip_set_device(dom.index) != GR4HIP_OK) return nullptr; // no such device: the edge falls back to the default resource
        return pinned_resource();
    });
}

// ---------------------------------------------------------------------------------------------- copy threads
// A pageable edge has to be staged through page-locked memory before the copy engine can take it, and one host thread's memcpy (10-15 GB/s)
// is far below what the link moves (DESIGN.md "Host feed"): large staging copies are cut into slices for a few helper threads.
// GR4HIP_COPY_THREADS sets the helper count (default: an eighth of the host's hardware threads, 1 .. 7; 0: the calling thread copies alone).
class CopyPool {
    struct Slice { char* d; const char* s; std::size_t n; std::size_t* pending; }; // pending: the slice counter of the copy() call it belongs to
    std::vector<std::thread>  _threads;
    std::mutex                _m;
    std::condition_variable   _cv, _cv_done;
    std::vector<Slice>        _work;
    bool                      _stop    = false;
    void run() {
        for (;;) {
            Slice sl;
            {
                std::unique_lock lk(_m);
                _cv.wait(lk, [&] { return _stop || !_work.empty(); });
                if (_stop) return;
                sl = _work.back();
                _work.pop_back();
            }
            std::memcpy(sl.d, sl.s, sl.n);
            std::lock_guard lk(_m);
            if (--*sl.pending == 0) _cv_done.notify_all();
        }
    }
    CopyPool() {
        const char* e = std::getenv("GR4HIP_COPY_THREADS");
        const long  k = e ? std::strtol(e, nullptr, 10) : std::clamp<long>(static_cast<long>(std::thread::hardware_concurrency()) / 8, 1, 7); // (7 on the 256-core boxes)
        for (long i = 0; i < std::clamp(k, 0L, 16L); ++i) _threads.emplace_back([this] { run(); });
    }
    ~CopyPool() {
        { std::lock_guard lk(_m); _stop = true; }
        _cv.notify_all();
        for (auto& t : _threads) t.join();
    }

public:
    static CopyPool& instance() { static CopyPool p; return p; }
    [[nodiscard]] std::size_t helpers() const { return _threads.size(); }
    void copy(void* dst, const void* src, std::size_t bytes) {
        constexpr std::size_t kMin = std::size_t(1) << 20; // below this a slice is not worth a wake-up
        const std::size_t parts = std::min(_threads.size() + 1, bytes / kMin);
        if (parts < 2) { std::memcpy(dst, src, bytes); return; }
        const std::size_t per = (bytes / parts + 63) & ~std::size_t(63);
        std::size_t       pending = parts - 1; // this call's own slices: concurrent callers (several scheduler threads, several graphs) do not wait for each other's
        {
            std::lock_guard lk(_m);
            for (std::size_t i = 1; i < parts; ++i) {
                const std::size_t at = i * per;
                _work.push_back({static_cast<char*>(dst) + at, static_cast<const char*>(src) + at, std::min(per, bytes - at), &pending});
            }
        }
        _cv.notify_all();
        std::memcpy(dst, src, per); // the caller's share
        std::unique_lock lk(_m);
        _cv_done.wait(lk, [&] { return pending == 0; });
    }
};

// ---------------------------------------------------------------------------------------------- stages
// the hooks of a gr4hip_fir handle follow `ab` (gr4hip_fir_set_prologue / _epilogue: the library copies the programs)
inline void apply_fir_hooks(gr4hip_fir_t* h, const AbsorbedPrograms& ab) {
    for (int pro = 1; pro >= 0; --pro) {
        const EwiseProgram& p    = pro ? ab.pre : ab.post;
        gr4hip_ewise_t*     prog = p.empty() ? nullptr : p.make();
        const int           rc   = pro ? gr4hip_fir_set_prologue(h, prog) : gr4hip_fir_set_epilogue(h, prog);
        if (prog) gr4hip_ewise_destroy(prog);
        check(rc, pro ? "gr4hip_fir_set_prologue" : "gr4hip_fir_set_epilogue");
    }
}

template <typename T>
struct FirStage final : Stage {
    gr4hip_fir_t* h = nullptr;
    std::vector<float> taps;
    std::size_t        decim = 1;
    AbsorbedPrograms   absorbed;
    std::string        _kind;
    static constexpr int kDtype = gr::detail::is_complex<T>::value ? GR4HIP_C32 : GR4HIP_F32;
    template <typename Taps>
    explicit FirStage(const Taps& b, std::size_t decimate = 1) : taps(b.begin(), b.end()), decim(std::max<std::size_t>(1, decimate)) {
        in_bytes = out_bytes = sizeof(T);
        in_chunk = decim;
        check(gr4hip_fir_create(&h, kDtype, taps.data(), taps.size(), decim), "gr4hip_fir_create");
        check(gr4hip_fir_set_guard_mode(h, options().guard_mode), "gr4hip_fir_set_guard_mode");
        name();
    }
    ~FirStage() override { gr4hip_fir_destroy(h); }
    void name() { _kind = std::string(gr::detail::is_complex<T>::value ? "fir_c32" : "fir_f32") + (decim > 1 ? "/" + std::to_string(decim) : std::string()) + absorbed.suffix(); }
    // the neighbours' per-sample ops in this filter's launch: gains fold into the taps, the rest are load / store hooks (include/gr4hip.h)
    bool absorb(const EwiseProgram& p, bool before) override {
        if (!absorbed.take(p, before, kDtype)) return false;
        apply_fir_hooks(h, absorbed);
        name();
        return true;
    }
    void clear_absorbed() override {
        if (absorbed.pre.empty() && absorbed.post.empty()) return;
        absorbed.clear();
        apply_fir_hooks(h, absorbed);
        name();
    }
    template <typename Taps>
    void set_taps(const Taps& b) { // fir_filter::settingsChanged (time_domain_filter.hpp:38-42): new taps, the history survives
        taps.assign(b.begin(), b.end());
        check(gr4hip_fir_set_taps(h, taps.data(), taps.size()), "gr4hip_fir_set_taps");
    }
    std::string_view kind() const override { return _kind; }
    int enqueue(const void* in, std::size_t n, void* out, std::size_t* n_out, gr4hip_stream_t s) override { return gr4hip_fir_process(h, in, n, out, n_out, s); }
};

struct PowerSpectrumStage final : Stage {
    gr4hip_fft_t* h = nullptr;
    std::size_t   N;
    int           window;
    EwiseProgram  post; // float blocks behind |X|^2 (normalisation, an offset): in the transform's launch (gr4hip_fft_set_epilogue)
    std::string   _kind = "power_spectrum_c32";
    PowerSpectrumStage(std::size_t fftSize, int win) : N(fftSize), window(win) {
        in_bytes = 8; out_bytes = 4; in_chunk = out_chunk = fftSize;
        post.dtype = GR4HIP_F32;
        check(gr4hip_fft_create(&h, GR4HIP_C32, fftSize, win, 0), "gr4hip_fft_create");
    }
    ~PowerSpectrumStage() override { gr4hip_fft_destroy(h); }
    std::string_view kind() const override { return _kind; }
    void apply() {
        gr4hip_ewise_t* prog = post.empty() ? nullptr : post.make();
        const int       rc   = gr4hip_fft_set_epilogue(h, prog);
        if (prog) gr4hip_ewise_destroy(prog);
        check(rc, "gr4hip_fft_set_epilogue");
        _kind = post.empty() ? "power_spectrum_c32" : "power_spectrum_c32[post: " + post.describe() + "]";
    }
    bool absorb(const EwiseProgram& p, bool before) override {
        if (before || p.dtype != GR4HIP_F32) return false;
        post.append(p);
        apply();
        return true;
    }
    void clear_absorbed() override { if (!post.empty()) { post.ops.clear(); apply(); } }
    int enqueue(const void* in, std::size_t n, void* out, std::size_t* n_out, gr4hip_stream_t s) override {
        *n_out = (n / N) * N;
        return gr4hip_fft_mag2(h, in, n / N, static_cast<float*>(out), s);
    }
};

// fir_filter<complex<float>> -> PowerSpectrum fused into one launch
struct ChainStage final : Stage {
    gr4hip_chain_t* h = nullptr;
    std::size_t     N;
    ChainStage(const std::vector<float>& taps, std::size_t fftSize, int window) : N(fftSize) {
        in_bytes = 8; out_bytes = 4; in_chunk = out_chunk = fftSize;
        check(gr4hip_chain_create(&h, taps.data(), taps.size(), fftSize, window, GR4HIP_CHAIN_AUTO), "gr4hip_chain_create");
        check(gr4hip_chain_set_guard_mode(h, options().guard_mode), "gr4hip_chain_set_guard_mode");
    }
    ~ChainStage() override { gr4hip_chain_destroy(h); }
    std::string_view kind() const override { return "chain_fir_fft_mag2"; }
    int algo() const { int a = 0; gr4hip_chain_get_algo(h, &a); return a; }
    int enqueue(const void* in, std::size_t n, void* out, std::size_t* n_out, gr4hip_stream_t s) override {
        std::size_t frames = 0;
        const int   rc = gr4hip_chain_process(h, in, n, static_cast<float*>(out), &frames, s);
        *n_out = frames * N;
        return rc;
    }
};

template <typename T>
constexpr int dtype_of() { // gr4hip_dtype of a sample type
    if constexpr (std::is_same_v<T, std::uint8_t>) return GR4HIP_U8; else if constexpr (std::is_same_v<T, std::uint16_t>) return GR4HIP_U16;
    else if constexpr (std::is_same_v<T, std::uint32_t>) return GR4HIP_U32; else if constexpr (std::is_same_v<T, std::uint64_t>) return GR4HIP_U64;
    else if constexpr (std::is_same_v<T, std::int8_t>) return GR4HIP_I8; else if constexpr (std::is_same_v<T, std::int16_t>) return GR4HIP_I16;
    else if constexpr (std::is_same_v<T, std::int32_t>) return GR4HIP_I32; else if constexpr (std::is_same_v<T, std::int64_t>) return GR4HIP_I64;
    else if constexpr (std::is_same_v<T, float>) return GR4HIP_F32; else if constexpr (std::is_same_v<T, double>) return GR4HIP_F64;
    else if constexpr (std::is_same_v<T, std::complex<float>>) return GR4HIP_C32; else if constexpr (std::is_same_v<T, std::complex<double>>) return GR4HIP_C64;
    else if constexpr (std::is_same_v<T, gr::UncertainValue<float>>) return GR4HIP_UF32; // {value, uncertainty}: the struct's layout is the device's element
    else { static_assert(std::is_same_v<T, gr::UncertainValue<double>>, "no gr4hip_dtype for this sample type"); return GR4HIP_UF64; }
}

template <typename T, int OP>
struct MathConstStage final : Stage {
    T            value;
    EwiseProgram prog;
    explicit MathConstStage(T v) : value(v) {
        in_bytes = out_bytes = sizeof(T);
        prog.dtype = dtype();
        EwiseProgram::Op op;
        op.kind = OP;
        std::memcpy(op.value.data(), &value, sizeof(T));
        prog.ops.push_back(op);
    }
    std::string_view kind() const override { return "math_const"; }
    static constexpr int dtype() { return dtype_of<T>(); }
    const EwiseProgram*  program() const override { // (UncertainValue elements: one launch per block, gr4hip_ewise_create refuses them)
        if constexpr (gr::UncertainValueLike<T>) return nullptr; else return &prog;
    }
    int enqueue(const void* in, std::size_t n, void* out, std::size_t* n_out, gr4hip_stream_t s) override {
        *n_out = n;
        return gr4hip_math_const(OP, dtype(), in, out, n, &value, s);
    }
};

// a run of per-sample blocks: ONE launch, the values in registers between the ops (gr4hip_ewise_process)
struct EwiseStage final : Stage {
    EwiseProgram    prog;
    gr4hip_ewise_t* h = nullptr;
    std::string     _kind;
    explicit EwiseStage(EwiseProgram p) : prog(std::move(p)) {
        static constexpr std::size_t bytes[12] = {1, 2, 4, 8, 1, 2, 4, 8, 4, 8, 8, 16};
        in_bytes = out_bytes = bytes[std::clamp(prog.dtype, 0, 11)];
        rebuild();
    }
    ~EwiseStage() override { if (h) gr4hip_ewise_destroy(h); }
    void rebuild() {
        gr4hip_ewise_t* fresh = prog.make();
        if (h) gr4hip_ewise_destroy(h);
        h     = fresh;
        _kind = "ewise[" + prog.describe() + "]";
    }
    std::string_view    kind() const override { return _kind; }
    const EwiseProgram* program() const override { return &prog; }
    bool absorb(const EwiseProgram& p, bool before) override { // program + program = program
        if (p.dtype != prog.dtype) return false;
        if (before) prog.prepend(p); else prog.append(p);
        rebuild();
        return true;
    }
    int enqueue(const void* in, std::size_t n, void* out, std::size_t* n_out, gr4hip_stream_t s) override {
        *n_out = n;
        return gr4hip_ewise_process(h, in, out, n, s);
    }
};

// iir_filter<float, form> (one section with the user's b, a) and designed cascades
struct IirStage final : Stage {
    gr4hip_iir_t*      h = nullptr;
    int                form;
    std::size_t        nsections, nb, na;
    std::vector<float> b, a;
    double             gain = 1.0; // neighbouring gains folded into the first section's numerator (the cascade is linear)
    std::string        _kind = "iir_f32";
    IirStage(int form_, std::size_t nsections_, const std::vector<float>& b_, std::size_t nb_, const std::vector<float>& a_, std::size_t na_)
        : form(form_), nsections(nsections_), nb(nb_), na(na_), b(b_), a(a_) {
        check(gr4hip_iir_create(&h, form, nsections, b.data(), nb, a.data(), na), "gr4hip_iir_create");
    }
    ~IirStage() override { gr4hip_iir_destroy(h); }
    void regain(double g) { // a new handle with the first section's numerator scaled (plan time: no state yet; a live stage restarts from zero state)
        std::vector<float> bs = b;
        for (std::size_t k = 0; k < nb; ++k) bs[k] = static_cast<float>(g * double(b[k]));
        gr4hip_iir_t* fresh = nullptr;
        check(gr4hip_iir_create(&fresh, form, nsections, bs.data(), nb, a.data(), na), "gr4hip_iir_create");
        gr4hip_iir_destroy(h);
        h     = fresh;
        gain  = g;
        _kind = g == 1.0 ? "iir_f32" : "iir_f32[gain folded]";
    }
    bool absorb(const EwiseProgram& p, bool /*before*/) override { // a gain commutes with the filter: in front or behind, it scales the numerator
        double g = 1.0;
        if (p.dtype != GR4HIP_F32 || !AbsorbedPrograms::real_gain(p, &g)) return false;
        regain(gain * g);
        return true;
    }
    void clear_absorbed() override { if (gain != 1.0) regain(1.0); }
    std::string_view kind() const override { return _kind; }
    int enqueue(const void* in, std::size_t n, void* out, std::size_t* n_out, gr4hip_stream_t s) override {
        *n_out = n;
        return gr4hip_iir_process(h, static_cast<const float*>(in), n, static_cast<float*>(out), s);
    }
};

template <typename T>
struct DecimatorStage final : Stage {
    std::size_t      decim;
    EwiseProgram     absorbed; // per-sample blocks around the decimator, all applied BEHIND it (memoryless blocks commute with dropping samples): one launch
    gr4hip_ewise_t*  h = nullptr;
    std::string      _kind = "decimator";
    explicit DecimatorStage(std::size_t d) : decim(std::max<std::size_t>(1, d)) { in_bytes = out_bytes = sizeof(T); in_chunk = decim; out_chunk = 1; absorbed.dtype = dtype_of<T>(); }
    ~DecimatorStage() override { if (h) gr4hip_ewise_destroy(h); }
    std::string_view kind() const override { return _kind; }
    void refresh() {
        if (h) gr4hip_ewise_destroy(h);
        h     = absorbed.empty() ? nullptr : absorbed.make();
        _kind = absorbed.empty() ? "decimator" : "decimator[post: " + absorbed.describe() + "]";
    }
    bool absorb(const EwiseProgram& p, bool before) override {
        if (p.dtype != dtype_of<T>()) return false;
        for (const auto& o : p.ops)
            if (before && o.kind == EwiseProgram::kRotator) return false; // a rotator's phase counts samples: it does not commute with dropping them
        if (before) absorbed.prepend(p); else absorbed.append(p);
        refresh();
        return true;
    }
    void clear_absorbed() override { if (!absorbed.empty()) { absorbed.ops.clear(); refresh(); } }
    int enqueue(const void* in, std::size_t n, void* out, std::size_t* n_out, gr4hip_stream_t s) override {
        if (h) return gr4hip_ewise_decimate(h, in, n, decim, out, n_out, s);
        return gr4hip_decimate(dtype_of<T>(), in, n, decim, out, n_out, s);
    }
};

// interpolating FIR: polyphase kernel, n_out = n L
template <typename T>
struct InterpStage final : Stage {
    gr4hip_fir_interp_t* h = nullptr;
    std::size_t          L;
    std::vector<float>   taps;
    double               gain = 1.0; // neighbouring gains folded into the taps (linear)
    InterpStage(const std::vector<float>& b, std::size_t interp) : L(std::max<std::size_t>(1, interp)), taps(b) {
        in_bytes = out_bytes = sizeof(T); in_chunk = 1; out_chunk = L;
        check(gr4hip_fir_interp_create(&h, dtype_of<T>(), b.data(), b.size(), L), "gr4hip_fir_interp_create");
    }
    ~InterpStage() override { gr4hip_fir_interp_destroy(h); }
    void regain(double g) {
        std::vector<float> bs(taps.size());
        for (std::size_t k = 0; k < taps.size(); ++k) bs[k] = static_cast<float>(g * double(taps[k]));
        check(gr4hip_fir_interp_set_taps(h, bs.data(), bs.size()), "gr4hip_fir_interp_set_taps"); // (the history is kept)
        gain = g;
    }
    bool absorb(const EwiseProgram& p, bool /*before*/) override {
        double g = 1.0;
        if (p.dtype != dtype_of<T>() || !AbsorbedPrograms::real_gain(p, &g)) return false;
        regain(gain * g);
        return true;
    }
    void clear_absorbed() override { if (gain != 1.0) regain(1.0); }
    std::string_view kind() const override { return gain == 1.0 ? "fir_interp" : "fir_interp[gain folded]"; }
    int enqueue(const void* in, std::size_t n, void* out, std::size_t* n_out, gr4hip_stream_t s) override { return gr4hip_fir_interp_process(h, in, n, out, n_out, s); }
};

struct RotatorStage final : Stage {
    gr4hip_rotator_t* h = nullptr;
    EwiseProgram      prog; // the closed-form rotator as a per-sample op (empty with the reference's recurrence: that one is sequential and stays a stage of its own)
    RotatorStage(float phase_increment, float initial_phase) {
        in_bytes = out_bytes = 8;
        check(gr4hip_rotator_create(&h, phase_increment, initial_phase), "gr4hip_rotator_create");
        if (options().rotator_reference_recurrence) check(gr4hip_rotator_set_algo(h, GR4HIP_ROTATOR_RECURRENCE), "gr4hip_rotator_set_algo");
        else if (std::isfinite(phase_increment) && std::isfinite(initial_phase)) {
            prog.dtype = GR4HIP_C32;
            EwiseProgram::Op op;
            op.kind = EwiseProgram::kRotator;
            op.inc  = phase_increment;
            op.ph0  = initial_phase;
            prog.ops.push_back(op);
        }
    }
    ~RotatorStage() override { gr4hip_rotator_destroy(h); }
    const EwiseProgram* program() const override { return prog.empty() ? nullptr : &prog; }
    std::string_view kind() const override { return "rotator_c32"; }
    int enqueue(const void* in, std::size_t n, void* out, std::size_t* n_out, gr4hip_stream_t s) override {
        *n_out = n;
        return gr4hip_rotator_process(h, in, out, n, s);
    }
};

// ---- the float64 instantiations the reference registers (time_domain_filter.hpp:20, 57-60; Rotator.hpp:15; fourier/fft.hpp:29): plain FP64 kernels
struct Fir64Stage final : Stage {
    gr4hip_fir64_t* h = nullptr;
    explicit Fir64Stage(const std::vector<double>& b) {
        in_bytes = out_bytes = sizeof(double);
        check(gr4hip_fir64_create(&h, b.data(), b.size(), 1), "gr4hip_fir64_create");
    }
    ~Fir64Stage() override { gr4hip_fir64_destroy(h); }
    void set_taps(const std::vector<double>& b) { check(gr4hip_fir64_set_taps(h, b.data(), b.size()), "gr4hip_fir64_set_taps"); }
    std::string_view kind() const override { return "fir_f64"; }
    int enqueue(const void* in, std::size_t n, void* out, std::size_t* n_out, gr4hip_stream_t s) override {
        return gr4hip_fir64_process(h, static_cast<const double*>(in), n, static_cast<double*>(out), n_out, s);
    }
};
struct Iir64Stage final : Stage {
    gr4hip_iir64_t* h = nullptr;
    Iir64Stage(int form, const std::vector<double>& b, const std::vector<double>& a) {
        in_bytes = out_bytes = sizeof(double);
        check(gr4hip_iir64_create(&h, form, 1, b.data(), b.size(), a.data(), a.size()), "gr4hip_iir64_create");
    }
    ~Iir64Stage() override { gr4hip_iir64_destroy(h); }
    std::string_view kind() const override { return "iir_f64"; }
    int enqueue(const void* in, std::size_t n, void* out, std::size_t* n_out, gr4hip_stream_t s) override {
        *n_out = n;
        return gr4hip_iir64_process(h, static_cast<const double*>(in), n, static_cast<double*>(out), s);
    }
};
struct Rotator64Stage final : Stage {
    gr4hip_rotator64_t* h = nullptr;
    Rotator64Stage(double phase_increment, double initial_phase) {
        in_bytes = out_bytes = 16;
        check(gr4hip_rotator64_create(&h, phase_increment, initial_phase), "gr4hip_rotator64_create");
    }
    ~Rotator64Stage() override { gr4hip_rotator64_destroy(h); }
    std::string_view kind() const override { return "rotator_c64"; }
    int enqueue(const void* in, std::size_t n, void* out, std::size_t* n_out, gr4hip_stream_t s) override {
        *n_out = n;
        return gr4hip_rotator64_process(h, in, out, n, s);
    }
};

// BasicFilterProto<float, ...>: designed FIR (polyphase when decimating: only the kept outputs are computed) or designed IIR cascade at the
// full rate followed by the keep-every-D-th step (time_domain_filter.hpp:190-204)
struct BasicFilterStage final : Stage {
    gr4hip_fir_t*    fir = nullptr;
    gr4hip_iir_t*    iir = nullptr;
    std::size_t      decim;
    int              dtype; // GR4HIP_F32 as registered upstream, or GR4HIP_C32 (complex data, the designed real taps: FIR designs only)
    DevBuf           tmp;
    AbsorbedPrograms absorbed;
    std::string      _kind;
    BasicFilterStage(const gr::filter::DesignedFilter& d, std::size_t decimate, int dtype_ = GR4HIP_F32) : decim(std::max<std::size_t>(1, decimate)), dtype(dtype_) {
        in_chunk = decim; out_chunk = 1;
        in_bytes = out_bytes = dtype == GR4HIP_C32 ? 8 : 4;
        if (d.fir) {
            check(gr4hip_fir_create(&fir, dtype, d.taps.data(), d.taps.size(), decim), "gr4hip_fir_create");
        } else {
            if (dtype != GR4HIP_F32) throw std::invalid_argument("BasicFilter<complex<float>> on the device: FIR designs only");
            std::vector<float> b, a;
            for (std::size_t s = 0; s < d.b.size(); ++s) { b.insert(b.end(), d.b[s].begin(), d.b[s].end()); a.insert(a.end(), d.a[s].begin(), d.a[s].end()); }
            check(gr4hip_iir_create(&iir, GR4HIP_DF_II, d.b.size(), b.data(), 3, a.data(), 3), "gr4hip_iir_create");
        }
        name();
    }
    ~BasicFilterStage() override { if (fir) gr4hip_fir_destroy(fir); if (iir) gr4hip_iir_destroy(iir); }
    void name() { _kind = std::string(fir ? (decim > 1 ? "basic_fir_decim" : "basic_fir") : (decim > 1 ? "basic_iir_decim" : "basic_iir")) + absorbed.suffix(); }
    bool absorb(const EwiseProgram& p, bool before) override { // designed FIR: the neighbours ride in the filter's launch (gains in the taps, the rest as hooks)
        if (!fir || !absorbed.take(p, before, dtype)) return false;
        apply_fir_hooks(fir, absorbed);
        name();
        return true;
    }
    void clear_absorbed() override {
        if (!fir || (absorbed.pre.empty() && absorbed.post.empty())) return;
        absorbed.clear();
        apply_fir_hooks(fir, absorbed);
        name();
    }
    std::string_view kind() const override { return _kind; }
    int enqueue(const void* in, std::size_t n, void* out, std::size_t* n_out, gr4hip_stream_t s) override {
        if (fir) return gr4hip_fir_process(fir, in, n, out, n_out, s);
        if (decim == 1) { *n_out = n; return gr4hip_iir_process(iir, static_cast<const float*>(in), n, static_cast<float*>(out), s); }
        float* y = static_cast<float*>(tmp.ensure(n * sizeof(float)));
        if (const int rc = gr4hip_iir_process(iir, static_cast<const float*>(in), n, y, s)) return rc;
        return gr4hip_decimate(GR4HIP_F32, y, n, decim, out, n_out, s);
    }
};

template <typename op, typename T> constexpr int op_id() {
    if constexpr (std::is_same_v<op, std::plus<T>>) return GR4HIP_ADD; else if constexpr (std::is_same_v<op, std::minus<T>>) return GR4HIP_SUB;
    else if constexpr (std::is_same_v<op, std::multiplies<T>>) return GR4HIP_MUL; else return GR4HIP_DIV;
}

inline int window_id(const std::string& w) { // gr::algorithm::window::TypeNames (window.hpp:22-40), case-insensitive like magic_enum::enum_cast
    static constexpr const char* names[] = {"none", "rectangular", "hamming", "hann", "hannexp", "blackman", "nuttall", "blackmanharris", "blackmannuttall",
                                            "flattop", "exponential", "kaiser"};
    std::string lw(w);
    for (auto& c : lw) c = static_cast<char>(std::tolower(static_cast<unsigned char>(c)));
    for (int i = 0; i < 12; ++i)
        if (lw == names[i]) return i; // == GR4HIP_WIN_*
    throw std::invalid_argument("unknown window '" + w + "'");
}

// ---------------------------------------------------------------------------------------------- per-block offload at the seam
struct Offload { // state behind Block::_device_state (owned by the block through a shared_ptr<void> with this type's deleter)
    virtual ~Offload() = default;
    std::unique_ptr<Stage> stage;
    DevBuf                 d_in, d_out, h_in{true}, h_out{true};
    int                    device = 0;
    std::size_t            settings_generation = 0; // Block::_settings_generation the stage was built (or refreshed) for
};
// the block's Offload (created on first use, on the block's device); every call makes the block's device the calling thread's current device:
// one scheduler thread drives blocks of several "gpu:hip:<i>" domains, and allocations, attributes and launches follow the CURRENT device
template <typename State = Offload, typename BlockT>
State* offload_state(BlockT& blk) {
    auto* st = static_cast<State*>(static_cast<Offload*>(blk._device_state.get()));
    if (!st) {
        check(gr4hip_set_device(blk._domain.index), "gr4hip_set_device");
        st         = new State();
        st->device = blk._domain.index;
        blk._device_state = std::shared_ptr<void>(static_cast<Offload*>(st), [](void* p) { delete static_cast<Offload*>(p); });
    }
    check(gr4hip_set_device(st->device), "gr4hip_set_device");
    return st;
}

// make(blk): a fresh stage from the block's CURRENT settings.  update(stage, blk) (optional): bring a live stage up to date and return true when that
// is possible without losing its state (fir_filter keeps its history across new taps); otherwise the stage is rebuilt.  A settings change is seen
// through Block::_settings_generation (applySettings and settings-by-tag bump it): a stage is never used with stale taps or constants.
template <typename BlockT, typename MakeStage, typename UpdateStage = std::nullptr_t>
work::Status offload_work(BlockT& blk, std::size_t nIn, std::size_t nOut, MakeStage&& make, UpdateStage&& update = nullptr) {
    try {
        Offload* st = offload_state(blk);
        if (!st->stage) {
            st->stage               = make(blk);
            st->settings_generation = blk._settings_generation;
        } else if (st->settings_generation != blk._settings_generation) {
            bool kept = false;
            if constexpr (!std::is_same_v<std::decay_t<UpdateStage>, std::nullptr_t>) kept = update(*st->stage, blk);
            if (!kept) st->stage = make(blk);
            st->settings_generation = blk._settings_generation;
        }
        using TIn  = typename std::decay_t<decltype(blk.in)>::value_type;
        using TOut = typename std::decay_t<decltype(blk.out)>::value_type;
        const auto is = blk.in.buffer->read_span(nIn);
        auto       os = blk.out.buffer->write_span(nOut);
        // edges from the "hip" provider are page-locked: the copy engine takes them in place; ordinary edges are staged through page-locked buffers, large
        // spans by the copy threads (one thread's memcpy was the whole cost of this path: 0.3 Gsamples/s for a two-block chain)
        const bool  in_locked  = blk.in.buffer->resource() == pinned_resource(), out_locked = blk.out.buffer->resource() == pinned_resource();
        const void* src        = is.data();
        if (!in_locked) {
            CopyPool::instance().copy(st->h_in.ensure(nIn * sizeof(TIn)), is.data(), nIn * sizeof(TIn));
            src = st->h_in.p;
        }
        check(gr4hip_memcpy_h2d(st->d_in.ensure(nIn * sizeof(TIn)), src, nIn * sizeof(TIn), nullptr), "h2d");
        std::size_t produced = 0;
        check(st->stage->enqueue(st->d_in.p, nIn, st->d_out.ensure(nOut * sizeof(TOut)), &produced, nullptr), "kernel");
        if (produced != nOut) throw std::runtime_error("device stage produced an unexpected number of samples");
        void* dst = out_locked ? static_cast<void*>(os.data()) : st->h_out.ensure(nOut * sizeof(TOut));
        check(gr4hip_memcpy_d2h(dst, st->d_out.p, nOut * sizeof(TOut), nullptr), "d2h");
        check(gr4hip_stream_synchronize(nullptr), "sync");
        if (!out_locked) CopyPool::instance().copy(os.data(), st->h_out.p, nOut * sizeof(TOut));
        return work::Status::OK;
    } catch (const std::exception& e) {
        blk._log(std::string("device block '") + blk.name + "' failed: " + e.what());
        return work::Status::ERROR; // never a silent host fallback
    }
}

template <typename BlockT>
void release(BlockT& blk) { // early release (the state also goes with the block)
    blk._device_state.reset();
}

template <typename T>
requires(std::is_same_v<T, float> || std::is_same_v<T, std::complex<float>>)
struct Kernel<gr::filter::fir_filter<T>> {
    static std::unique_ptr<Stage> make_stage(gr::filter::fir_filter<T>& b) { return std::make_unique<FirStage<T>>(b.b); }
    static work::Status           work(gr::filter::fir_filter<T>& b, std::size_t nIn, std::size_t nOut) {
        return offload_work(b, nIn, nOut, make_stage, [](Stage& st, gr::filter::fir_filter<T>& blk) { static_cast<FirStage<T>&>(st).set_taps(blk.b); return true; });
    }
};
template <>
struct Kernel<gr::filter::fir_filter<double>> {
    using B = gr::filter::fir_filter<double>;
    static std::unique_ptr<Stage> make_stage(B& b) { return std::make_unique<Fir64Stage>(b.b); }
    static work::Status           work(B& b, std::size_t nIn, std::size_t nOut) {
        return offload_work(b, nIn, nOut, make_stage, [](Stage& st, B& blk) { static_cast<Fir64Stage&>(st).set_taps(blk.b); return true; });
    }
};
template <gr::filter::IIRForm form>
struct Kernel<gr::filter::iir_filter<double, form>> {
    using B = gr::filter::iir_filter<double, form>;
    static std::unique_ptr<Stage> make_stage(B& b) { return std::make_unique<Iir64Stage>(static_cast<int>(form), b.b, b.a); }
    static work::Status           work(B& b, std::size_t nIn, std::size_t nOut) { return offload_work(b, nIn, nOut, make_stage); }
};
template <>
struct Kernel<gr::blocks::math::Rotator<std::complex<double>>> {
    using B = gr::blocks::math::Rotator<std::complex<double>>;
    static std::unique_ptr<Stage> make_stage(B& b) { return std::make_unique<Rotator64Stage>(b.phase_increment, b._accumulated_phase); }
    static work::Status           work(B& b, std::size_t nIn, std::size_t nOut) { return offload_work(b, nIn, nOut, make_stage); }
};
template <typename T, typename op>
struct Kernel<gr::blocks::math::MathOpImpl<T, op>> {
    using B = gr::blocks::math::MathOpImpl<T, op>;
    static std::unique_ptr<Stage> make_stage(B& b) { return std::make_unique<MathConstStage<T, op_id<op, T>()>>(b.value); }
    static work::Status           work(B& b, std::size_t nIn, std::size_t nOut) { return offload_work(b, nIn, nOut, make_stage); }
};
template <>
struct Kernel<gr::blocks::fft::PowerSpectrum<std::complex<float>>> {
    using B = gr::blocks::fft::PowerSpectrum<std::complex<float>>;
    static std::unique_ptr<Stage> make_stage(B& b) { return std::make_unique<PowerSpectrumStage>(b.fftSize, window_id(b.window)); }
    static work::Status           work(B& b, std::size_t nIn, std::size_t nOut) { return offload_work(b, nIn, nOut, make_stage); }
};

template <gr::filter::IIRForm form>
struct Kernel<gr::filter::iir_filter<float, form>> {
    using B = gr::filter::iir_filter<float, form>;
    static std::unique_ptr<Stage> make_stage(B& b) { return std::make_unique<IirStage>(static_cast<int>(form), 1, b.b, b.b.size(), b.a, b.a.size()); }
    static work::Status           work(B& b, std::size_t nIn, std::size_t nOut) { return offload_work(b, nIn, nOut, make_stage); }
};
template <typename T>
struct Kernel<gr::filter::Decimator<T>> {
    using B = gr::filter::Decimator<T>;
    static std::unique_ptr<Stage> make_stage(B& b) { return std::make_unique<DecimatorStage<T>>(b.decim); }
    static work::Status           work(B& b, std::size_t nIn, std::size_t nOut) { return offload_work(b, nIn, nOut, make_stage); }
};
template <typename T>
    requires(std::is_same_v<T, float> || std::is_same_v<T, std::complex<float>>)
struct Kernel<gr::filter::fir_interpolator<T>> {
    using B = gr::filter::fir_interpolator<T>;
    static std::unique_ptr<Stage> make_stage(B& b) { return std::make_unique<InterpStage<T>>(b.b, b.interpolate); }
    static work::Status           work(B& b, std::size_t nIn, std::size_t nOut) { return offload_work(b, nIn, nOut, make_stage); }
};
template <typename T, typename... Args>
requires(std::is_same_v<T, float> || std::is_same_v<T, std::complex<float>>)
struct Kernel<gr::filter::BasicFilterProto<T, Args...>> {
    using B = gr::filter::BasicFilterProto<T, Args...>;
    static std::unique_ptr<Stage> make_stage(B& b) {
        if (!b._designed) b.designFilter();
        return std::make_unique<BasicFilterStage>(b._design, B::TParent::ResamplingControl::kIsConst ? 1 : b.decimate.value, dtype_of<T>());
    }
    static work::Status work(B& b, std::size_t nIn, std::size_t nOut) { return offload_work(b, nIn, nOut, make_stage); }
};
template <>
struct Kernel<gr::blocks::math::Rotator<std::complex<float>>> {
    using B = gr::blocks::math::Rotator<std::complex<float>>;
    static std::unique_ptr<Stage> make_stage(B& b) { return std::make_unique<RotatorStage>(b.phase_increment, b._accumulated_phase); }
    static work::Status           work(B& b, std::size_t nIn, std::size_t nOut) { return offload_work(b, nIn, nOut, make_stage); }
};

// ---- merged blocks (gr4/merge.hpp): the parts of a Merge<> become stages of one block; intermediates stay in HBM
struct SeqStage final : Stage {
    std::unique_ptr<Stage> a, b;
    DevBuf                 mid;
    std::string            _kind;
    SeqStage(std::unique_ptr<Stage> a_, std::unique_ptr<Stage> b_) : a(std::move(a_)), b(std::move(b_)), _kind(std::string(a->kind()) + " + " + std::string(b->kind())) {
        in_bytes = a->in_bytes; out_bytes = b->out_bytes;
        const std::size_t k = b->in_chunk / std::gcd(b->in_chunk, a->out_chunk); // A chunks so that B sees whole chunks
        in_chunk  = k * a->in_chunk;
        out_chunk = k * a->out_chunk / b->in_chunk * b->out_chunk;
    }
    std::string_view kind() const override { return _kind; }
    int enqueue(const void* in, std::size_t n, void* out, std::size_t* n_out, gr4hip_stream_t s) override {
        std::size_t m = 0;
        if (const int rc = a->enqueue(in, n, mid.ensure(std::max<std::size_t>(1, n / a->in_chunk * a->out_chunk) * a->out_bytes), &m, s)) return rc;
        return b->enqueue(mid.p, m, out, n_out, s);
    }
};
// ---- kernel-level fusion of adjacent stages: the run-time Merge<> (BlockMerging.hpp:126-240: "one processOne through both blocks, bypassing runtime buffers")
//   * per-sample stage + per-sample stage (MathOpImpl, closed-form Rotator)      -> ONE element-wise program: one launch, values in registers
//   * per-sample stage next to a filter that can take it (Stage::absorb)         -> the filter's launch: gains fold into its coefficients, anything else is its
//                                                                                   load / store hook.  Behind a filter first (its output rate is the lower one)
//   * fir_filter<T> -> Decimator<T>                                              -> the polyphase decimating FIR (only the kept outputs are computed)
//   * fir_filter<complex<float>> -> PowerSpectrum                                -> the fused FIR -> FFT -> |.|^2 kernel (gr4hip_chain_*)
// A group keeps the members it stands for; `anchor` is the member whose stage survived with the others absorbed into it (npos: the group's stage was made anew).
struct FusedGroup {
    std::unique_ptr<Stage>   stage;
    std::vector<std::size_t> members;
    std::size_t              anchor = static_cast<std::size_t>(-1);
};
namespace detail {
template <typename T>
inline std::unique_ptr<Stage> fuse_fir_decimator(Stage& a, Stage& b) {
    auto* fir = dynamic_cast<FirStage<T>*>(a.self());
    auto* dec = dynamic_cast<DecimatorStage<T>*>(b.self());
    if (!fir || !dec || fir->decim != 1 || dec->decim < 2 || !fir->absorbed.post.empty()) return nullptr;
    auto fused = std::make_unique<FirStage<T>>(fir->taps, dec->decim);
    if (!fir->absorbed.pre.empty() && !fused->absorb(fir->absorbed.pre, true)) return nullptr;
    if (!dec->absorbed.empty() && !fused->absorb(dec->absorbed, false)) return nullptr; // what the decimator had taken in rides behind the decimating filter
    return fused;
}
inline std::unique_ptr<Stage> fuse_pair(Stage& a, Stage& b); // defined below ChainStage's users
} // namespace detail
inline std::vector<FusedGroup> fuse_stages(std::vector<std::unique_ptr<Stage>> made) {
    std::vector<FusedGroup> g;
    for (std::size_t i = 0; i < made.size(); ++i) {
        FusedGroup f;
        f.stage   = std::move(made[i]);
        f.members = {i};
        f.anchor  = i;
        g.push_back(std::move(f));
    }
    const auto merge_into = [&](std::size_t keep, std::size_t drop, std::unique_ptr<Stage> replacement) { // groups keep and drop are neighbours
        FusedGroup& k = g[keep];
        FusedGroup& d = g[drop];
        if (replacement) { k.stage = std::move(replacement); k.anchor = static_cast<std::size_t>(-1); }
        if (drop < keep) k.members.insert(k.members.begin(), d.members.begin(), d.members.end());
        else k.members.insert(k.members.end(), d.members.begin(), d.members.end());
        g.erase(g.begin() + static_cast<std::ptrdiff_t>(drop));
    };
    for (bool changed = true; changed;) {
        changed = false;
        for (std::size_t i = 0; i + 1 < g.size() && !changed; ++i) {
            Stage&              a  = *g[i].stage;
            Stage&              b  = *g[i + 1].stage;
            const EwiseProgram* pa = a.program();
            const EwiseProgram* pb = b.program();
            if (pa && pb && pa->dtype == pb->dtype) { // program + program
                if (dynamic_cast<EwiseStage*>(a.self())) { a.absorb(*pb, false); merge_into(i, i + 1, nullptr); g[i].anchor = static_cast<std::size_t>(-1); }
                else if (dynamic_cast<EwiseStage*>(b.self())) { b.absorb(*pa, true); merge_into(i + 1, i, nullptr); g[i].anchor = static_cast<std::size_t>(-1); }
                else {
                    auto e = std::make_unique<EwiseStage>(*pa);
                    e->absorb(*pb, false);
                    merge_into(i, i + 1, std::move(e));
                }
                changed = true;
            } else if (pb && !pa && a.absorb(*pb, false)) { // behind a filter: its store hook
                merge_into(i, i + 1, nullptr);
                changed = true;
            } else if (pa && !pb && b.absorb(*pa, true)) { // in front of a filter: its load hook
                merge_into(i + 1, i, nullptr);
                changed = true;
            } else if (auto fused = detail::fuse_pair(a, b)) {
                merge_into(i, i + 1, std::move(fused));
                changed = true;
            }
        }
    }
    return g;
}
// the stages of `made` as ONE stage: fused where kernels exist for it, back to back with the intermediates in HBM where not
inline std::unique_ptr<Stage> fuse_to_one(std::vector<std::unique_ptr<Stage>> made) {
    auto                   groups = fuse_stages(std::move(made));
    std::unique_ptr<Stage> acc    = std::move(groups.front().stage);
    for (std::size_t i = 1; i < groups.size(); ++i) acc = std::make_unique<SeqStage>(std::move(acc), std::move(groups[i].stage));
    return acc;
}

template <typename A, fixed_string OutA, typename B, fixed_string InB>
requires requires(A& a, B& b) { Kernel<A>::make_stage(a); Kernel<B>::make_stage(b); }
struct Kernel<gr::Merge<A, OutA, B, InB>> {
    using M = gr::Merge<A, OutA, B, InB>;
    static std::unique_ptr<Stage> make_stage(M& m);
    static work::Status           work(M& m, std::size_t nIn, std::size_t nOut) { return offload_work(m, nIn, nOut, make_stage); }
};
// an adder whose second input is its own output through a constant gain c, one sample late: y[n] = x[n] + c y[n-1] -- a 1-pole IIR, run by the
// parallel-in-time scan kernel instead of one sample per scheduler cycle (FeedbackMerge, BlockMerging.hpp:600-800)
template <fixed_string FO, fixed_string BO, fixed_string FI>
struct Kernel<gr::FeedbackMerge<gr::Adder<float>, FO, gr::blocks::math::MultiplyConst<float>, BO, FI>> {
    using M = gr::FeedbackMerge<gr::Adder<float>, FO, gr::blocks::math::MultiplyConst<float>, BO, FI>;
    static std::unique_ptr<Stage> make_pole(float gain, float c) { // y[n] = gain x[n] + c y[n-1]
        auto st = std::make_unique<IirStage>(GR4HIP_DF_I, 1, std::vector<float>{gain}, 1, std::vector<float>{1.f, -c}, 2);
        return st;
    }
    static std::unique_ptr<Stage> make_stage(M& m) { return make_pole(1.f, m.feedback.value); }
    static float                  pole(M& m) { return m.feedback.value; }
    static work::Status           work(M& m, std::size_t nIn, std::size_t nOut) { return offload_work(m, nIn, nOut, make_stage); }
};
// SplitMergeCombine on the device: every path is a stage fed from the same device span, the signed outputs are folded with the math kernels
struct SplitStage final : Stage {
    std::vector<std::unique_ptr<Stage>> paths;
    std::vector<double>                 signs;
    std::vector<std::unique_ptr<DevBuf>> tmp;
    std::string                         _kind = "split(";
    SplitStage(std::vector<std::unique_ptr<Stage>> p, std::vector<double> s) : paths(std::move(p)), signs(std::move(s)) {
        for (std::size_t i = 0; i < paths.size(); ++i) {
            if (paths[i]->in_chunk != 1 || paths[i]->out_chunk != 1 || paths[i]->in_bytes != 4 || paths[i]->out_bytes != 4) throw std::invalid_argument("SplitMergeCombine on the device: float paths at one rate");
            tmp.push_back(std::make_unique<DevBuf>());
            _kind += std::string(i ? " | " : "") + std::string(paths[i]->kind());
        }
        _kind += ")";
    }
    std::string_view kind() const override { return _kind; }
    int enqueue(const void* in, std::size_t n, void* out, std::size_t* n_out, gr4hip_stream_t s) override {
        *n_out = n;
        for (std::size_t i = 0; i < paths.size(); ++i) {
            void*       dst = i == 0 ? out : tmp[i]->ensure(n * 4);
            std::size_t m   = 0;
            if (const int rc = paths[i]->enqueue(in, n, dst, &m, s)) return rc;
            if (m != n) return GR4HIP_ERROR;
            const float sg = static_cast<float>(signs[i]);
            if (i == 0) {
                if (sg != 1.f) { if (const int rc = gr4hip_math_const(GR4HIP_MUL, GR4HIP_F32, out, out, n, &sg, s)) return rc; }
            } else {
                const float mag = std::fabs(sg);
                if (mag != 1.f) { if (const int rc = gr4hip_math_const(GR4HIP_MUL, GR4HIP_F32, dst, dst, n, &mag, s)) return rc; }
                const void* ins[2] = {out, dst};
                if (const int rc = gr4hip_math_nary(sg < 0 ? GR4HIP_SUB : GR4HIP_ADD, GR4HIP_F32, ins, 2, out, n, s)) return rc;
            }
        }
        return GR4HIP_OK;
    }
};
template <typename SignsT, typename... Paths>
requires(std::is_same_v<gr::detail::in_type_t<std::tuple_element_t<0, std::tuple<Paths...>>>, float> && (requires(Paths& p) { Kernel<Paths>::make_stage(p); } && ...))
struct Kernel<gr::SplitMergeCombineImpl<SignsT, Paths...>> {
    using M = gr::SplitMergeCombineImpl<SignsT, Paths...>;
    static std::unique_ptr<Stage> make_stage(M& m) {
        std::vector<std::unique_ptr<Stage>> st;
        std::vector<double>                 sg;
        [&]<std::size_t... I>(std::index_sequence<I...>) {
            (st.push_back(Kernel<Paths>::make_stage(m.template path<I>())), ...);
            (sg.push_back(M::sign(I)), ...);
        }(std::index_sequence_for<Paths...>{});
        return std::make_unique<SplitStage>(std::move(st), std::move(sg));
    }
    static work::Status work(M& m, std::size_t nIn, std::size_t nOut) { return offload_work(m, nIn, nOut, make_stage); }
};
// the gain of a feedback path when the path is nothing but gains: a MultiplyConst, or a SplitMergeCombine of such paths (sum of the signed gains)
namespace detail {
inline bool loop_gain(gr::blocks::math::MultiplyConst<float>& b, float* g) { *g = b.value; return true; }
template <typename SignsT, typename... Paths>
bool loop_gain(gr::SplitMergeCombineImpl<SignsT, Paths...>& b, float* g);
template <std::size_t I, typename SM>
bool loop_gain_path(SM& b, float* sum) { // one path at a time (a fold over a lambda with `if constexpr (requires ...)` inside trips g++ 11)
    if constexpr (I == std::tuple_size_v<decltype(b._paths)>) {
        return true;
    } else {
        float gi = 0.f;
        if constexpr (requires { loop_gain(b.template path<I>(), &gi); }) {
            if (!loop_gain(b.template path<I>(), &gi)) return false;
            *sum += static_cast<float>(SM::sign(I)) * gi;
            return loop_gain_path<I + 1>(b, sum);
        } else {
            return false;
        }
    }
}
template <typename SignsT, typename... Paths>
bool loop_gain(gr::SplitMergeCombineImpl<SignsT, Paths...>& b, float* g) {
    *g = 0.f;
    return loop_gain_path<0>(b, g);
}
} // namespace detail
template <fixed_string FO, typename SignsT, typename... Paths, fixed_string BO, fixed_string FI>
requires requires(gr::SplitMergeCombineImpl<SignsT, Paths...>& b, float* g) { detail::loop_gain(b, g); }
struct Kernel<gr::FeedbackMerge<gr::Adder<float>, FO, gr::SplitMergeCombineImpl<SignsT, Paths...>, BO, FI>> {
    using M = gr::FeedbackMerge<gr::Adder<float>, FO, gr::SplitMergeCombineImpl<SignsT, Paths...>, BO, FI>;
    static std::unique_ptr<Stage> make_pole(float gain, float c) { return std::make_unique<IirStage>(GR4HIP_DF_I, 1, std::vector<float>{gain}, 1, std::vector<float>{1.f, -c}, 2); }
    static std::unique_ptr<Stage> make_stage(M& m) {
        float c = 0.f;
        if (!detail::loop_gain(m.feedback, &c)) throw std::invalid_argument("FeedbackMerge on the device: the feedback path must reduce to a constant gain");
        return make_pole(1.f, c);
    }
    static float        pole(M& m) { float c = 0.f; detail::loop_gain(m.feedback, &c); return c; }
    static work::Status work(M& m, std::size_t nIn, std::size_t nOut) { return offload_work(m, nIn, nOut, make_stage); }
};
namespace detail {
template <typename T>
struct is_pole_feedback : std::false_type {};
template <fixed_string FO, typename SignsT, typename... Paths, fixed_string BO, fixed_string FI>
struct is_pole_feedback<gr::FeedbackMerge<gr::Adder<float>, FO, gr::SplitMergeCombineImpl<SignsT, Paths...>, BO, FI>> : std::true_type {};
template <fixed_string FO, fixed_string BO, fixed_string FI>
struct is_pole_feedback<gr::FeedbackMerge<gr::Adder<float>, FO, gr::blocks::math::MultiplyConst<float>, BO, FI>> : std::true_type {};
} // namespace detail
template <typename A, fixed_string OutA, typename B, fixed_string InB>
requires requires(A& a, B& b) { Kernel<A>::make_stage(a); Kernel<B>::make_stage(b); }
std::unique_ptr<Stage> Kernel<gr::Merge<A, OutA, B, InB>>::make_stage(M& m) {
    // peephole: input gain -> pole feedback is still ONE first-order section (bm_MergeApi.cpp:59-60: y[n] = a x[n] + (1 - a) y[n-1])
    if constexpr (std::is_same_v<A, gr::blocks::math::MultiplyConst<float>> && detail::is_pole_feedback<B>::value) return Kernel<B>::make_pole(m.leftBlock.value, Kernel<B>::pole(m.rightBlock));
    else { // one launch where the parts fuse (per-sample parts into one program or into the neighbouring filter's launch), two with the intermediate in HBM where not
        std::vector<std::unique_ptr<Stage>> parts;
        parts.push_back(Kernel<A>::make_stage(m.leftBlock));
        parts.push_back(Kernel<B>::make_stage(m.rightBlock));
        return fuse_to_one(std::move(parts));
    }
}
namespace detail {
inline std::unique_ptr<Stage> fuse_pair(Stage& a, Stage& b) {
    if (auto f = fuse_fir_decimator<float>(a, b)) return f;
    if (auto f = fuse_fir_decimator<std::complex<float>>(a, b)) return f;
    auto* fir  = dynamic_cast<FirStage<std::complex<float>>*>(a.self());
    auto* spec = dynamic_cast<PowerSpectrumStage*>(b.self());
    if (fir && spec && fir->decim == 1 && fir->absorbed.post.empty() && spec->post.empty()) { // the fused FIR -> FFT -> |.|^2 kernel; a gain in front of the filter rides in its taps
        double g = 1.0;
        if (!fir->absorbed.pre.empty() && !AbsorbedPrograms::real_gain(fir->absorbed.pre, &g)) return nullptr;
        std::vector<float> taps(fir->taps.size());
        for (std::size_t k = 0; k < taps.size(); ++k) taps[k] = static_cast<float>(g * double(fir->taps[k]));
        return std::make_unique<ChainStage>(taps, spec->N, spec->window);
    }
    return nullptr;
}
} // namespace detail

// N inputs -> 1 output at the seam (Math.hpp:100-107): every input span goes to HBM, one fold kernel, one span back.  No Stage: a
// fan-in is not part of a linear device run, the planner leaves it to this per-block path.
template <typename T, typename op>
struct Kernel<gr::blocks::math::MathOpMultiPortImpl<T, op>> {
    using B = gr::blocks::math::MathOpMultiPortImpl<T, op>;
    struct State final : Offload {
        std::vector<std::unique_ptr<DevBuf>> d_ins;
    };
    static work::Status work(B& blk, std::size_t nIn, std::size_t nOut) {
        try {
            State* st = offload_state<State>(blk);
            while (st->d_ins.size() < blk.in.size()) st->d_ins.push_back(std::make_unique<DevBuf>());
            const std::size_t        bytes = nIn * sizeof(T);
            std::vector<const void*> ptrs;
            char*                    stage = static_cast<char*>(st->h_in.ensure(bytes * blk.in.size()));
            for (std::size_t i = 0; i < blk.in.size(); ++i) {
                std::memcpy(stage + i * bytes, blk.in[i].buffer->read_span(nIn).data(), bytes);
                check(gr4hip_memcpy_h2d(st->d_ins[i]->ensure(bytes), stage + i * bytes, bytes, nullptr), "h2d");
                ptrs.push_back(st->d_ins[i]->p);
            }
            check(gr4hip_math_nary(op_id<op, T>(), dtype_of<T>(), ptrs.data(), ptrs.size(), st->d_out.ensure(bytes), nIn, nullptr), "gr4hip_math_nary");
            check(gr4hip_memcpy_d2h(st->h_out.ensure(bytes), st->d_out.p, bytes, nullptr), "d2h");
            check(gr4hip_stream_synchronize(nullptr), "sync");
            std::memcpy(blk.out.buffer->write_span(nOut).data(), st->h_out.p, nOut * sizeof(T));
            return work::Status::OK;
        } catch (const std::exception& e) {
            blk._log(std::string("device block '") + blk.name + "' failed: " + e.what());
            return work::Status::ERROR;
        }
    }
};

// FFT block: nOut frames of fftSize samples in, nOut DataSets out.  The four signals and their ranges are computed on the device
// (gr4hip_fft_process); the host only assembles the descriptive part of the DataSet (axis, names, meta information).
template <typename T>
requires(std::is_same_v<T, float> || std::is_same_v<T, std::complex<float>>)
struct Kernel<gr::blocks::fft::FFT<T, DataSet<float>>> {
    using B = gr::blocks::fft::FFT<T, DataSet<float>>;
    struct State final : Offload {
        gr4hip_fft_t* h = nullptr;
        std::size_t   N = 0;
        std::string   window;
        int           flags = -1;
        DevBuf        d_sig, d_rng, h_rng{true};
        ~State() override { if (h) gr4hip_fft_destroy(h); }
    };
    static work::Status work(B& blk, std::size_t nIn, std::size_t nOut) {
        try {
            State* st = offload_state<State>(blk);
            const int flags = (blk.outputInDb ? GR4HIP_FFT_OUTPUT_IN_DB : 0) | (blk.outputInDeg ? GR4HIP_FFT_OUTPUT_IN_DEG : 0) | (blk.unwrapPhase ? GR4HIP_FFT_UNWRAP_PHASE : 0);
            if (!st->h || st->N != blk.fftSize || st->window != blk.window.value || st->flags != flags) { // settings changed: new plan
                if (st->h) gr4hip_fft_destroy(st->h);
                st->h = nullptr;
                check(gr4hip_fft_create(&st->h, dtype_of<T>(), blk.fftSize, window_id(blk.window), flags), "gr4hip_fft_create");
                st->N = blk.fftSize; st->window = blk.window; st->flags = flags;
            }
            const std::size_t N = st->N, M = blk.nBins(), frames = nOut;
            if (nIn != frames * N) throw std::runtime_error("FFT: the work loop must hand over whole frames");
            std::memcpy(st->h_in.ensure(nIn * sizeof(T)), blk.in.buffer->read_span(nIn).data(), nIn * sizeof(T));
            check(gr4hip_memcpy_h2d(st->d_in.ensure(nIn * sizeof(T)), st->h_in.p, nIn * sizeof(T), nullptr), "h2d");
            float* sig = static_cast<float*>(st->d_sig.ensure(4 * frames * M * sizeof(float))); // [mag | phase | re | im], each frames x M
            float* rng = static_cast<float*>(st->d_rng.ensure(frames * 8 * sizeof(float)));
            check(gr4hip_fft_process(st->h, st->d_in.p, frames, sig, sig + frames * M, sig + 2 * frames * M, sig + 3 * frames * M, rng, nullptr), "gr4hip_fft_process");
            check(gr4hip_memcpy_d2h(st->h_out.ensure(4 * frames * M * sizeof(float)), sig, 4 * frames * M * sizeof(float), nullptr), "d2h");
            check(gr4hip_memcpy_d2h(st->h_rng.ensure(frames * 8 * sizeof(float)), rng, frames * 8 * sizeof(float), nullptr), "d2h");
            check(gr4hip_stream_synchronize(nullptr), "sync");
            const float* hs = static_cast<const float*>(st->h_out.p);
            const float* hr = static_cast<const float*>(st->h_rng.p);
            auto         os = blk.out.buffer->write_span(nOut);
            const auto   skeleton = blk.datasetSkeleton();
            for (std::size_t f = 0; f < frames; ++f) {
                DataSet<float> ds = skeleton;
                for (std::size_t i = 0; i < 4; ++i) {
                    std::memcpy(ds.signalValues(i).data(), hs + (i * frames + f) * M, M * sizeof(float));
                    ds.signal_ranges[i] = {hr[f * 8 + 2 * i], hr[f * 8 + 2 * i + 1]};
                }
                os[f] = std::move(ds);
            }
            return work::Status::OK;
        } catch (const std::exception& e) {
            blk._log(std::string("device block '") + blk.name + "' failed: " + e.what());
            return work::Status::ERROR;
        }
    }
};

// FFT<double>: real double frames -> DataSet<double> (gr4hip_fft64_process; the per-signal ranges of fft.hpp:229-232 are taken on the host from the copied signals)
template <>
struct Kernel<gr::blocks::fft::FFT<double, DataSet<double>>> {
    using B = gr::blocks::fft::FFT<double, DataSet<double>>;
    struct State final : Offload {
        gr4hip_fft64_t* h = nullptr;
        std::size_t     N = 0;
        std::string     window;
        int             flags = -1;
        DevBuf          d_sig;
        ~State() override { if (h) gr4hip_fft64_destroy(h); }
    };
    static work::Status work(B& blk, std::size_t nIn, std::size_t nOut) {
        try {
            State* st = offload_state<State>(blk);
            const int flags = (blk.outputInDb ? GR4HIP_FFT_OUTPUT_IN_DB : 0) | (blk.outputInDeg ? GR4HIP_FFT_OUTPUT_IN_DEG : 0) | (blk.unwrapPhase ? GR4HIP_FFT_UNWRAP_PHASE : 0);
            if (!st->h || st->N != blk.fftSize || st->window != blk.window.value || st->flags != flags) {
                if (st->h) gr4hip_fft64_destroy(st->h);
                st->h = nullptr;
                check(gr4hip_fft64_create(&st->h, blk.fftSize, window_id(blk.window), flags), "gr4hip_fft64_create");
                st->N = blk.fftSize; st->window = blk.window; st->flags = flags;
            }
            const std::size_t N = st->N, M = blk.nBins(), frames = nOut;
            if (nIn != frames * N) throw std::runtime_error("FFT: the work loop must hand over whole frames");
            std::memcpy(st->h_in.ensure(nIn * sizeof(double)), blk.in.buffer->read_span(nIn).data(), nIn * sizeof(double));
            check(gr4hip_memcpy_h2d(st->d_in.ensure(nIn * sizeof(double)), st->h_in.p, nIn * sizeof(double), nullptr), "h2d");
            double* sig = static_cast<double*>(st->d_sig.ensure(4 * frames * M * sizeof(double))); // [mag | phase | re | im], each frames x M
            check(gr4hip_fft64_process(st->h, static_cast<const double*>(st->d_in.p), frames, sig, sig + frames * M, sig + 2 * frames * M, sig + 3 * frames * M, nullptr), "gr4hip_fft64_process");
            check(gr4hip_memcpy_d2h(st->h_out.ensure(4 * frames * M * sizeof(double)), sig, 4 * frames * M * sizeof(double), nullptr), "d2h");
            check(gr4hip_stream_synchronize(nullptr), "sync");
            const double* hs = static_cast<const double*>(st->h_out.p);
            auto          os = blk.out.buffer->write_span(nOut);
            const auto    skeleton = blk.datasetSkeleton();
            for (std::size_t f = 0; f < frames; ++f) {
                DataSet<double> ds = skeleton;
                for (std::size_t i = 0; i < 4; ++i) {
                    const double* v = hs + (i * frames + f) * M;
                    std::memcpy(ds.signalValues(i).data(), v, M * sizeof(double));
                    const auto [mn, mx] = std::minmax_element(v, v + M);
                    ds.signal_ranges[i] = {*mn, *mx};
                }
                os[f] = std::move(ds);
            }
            return work::Status::OK;
        } catch (const std::exception& e) {
            blk._log(std::string("device block '") + blk.name + "' failed: " + e.what());
            return work::Status::ERROR;
        }
    }
};

// ---------------------------------------------------------------------------------------------- GPU-resident BufferLike ring
// The device-side analogue of gr::CircularBuffer (core/include/gnuradio-4.0/CircularBuffer.hpp:191-236, BufferLike concept
// Buffer.hpp:73-103): one writer, any number of readers, capacity rounded up to the VMM granularity; the storage is
// mapped twice back to back (gr4hip_ring_*), so a span that wraps the physical end is still ONE contiguous device range and kernels
// never see a split.  Spans hold DEVICE pointers (usable as kernel arguments, not dereferenceable on the host).  Cursor protocol as
// in the reference: reserve -> (enqueue the producer's stream work) -> publish once its completion event has fired; get -> (enqueue
// the consumer) -> consume after completion.  Cursors are host-side atomics: the only cross-thread state, like gr::Sequence.
template <typename T>
class CircularBuffer {
    struct State {
        gr4hip_ring_t*                          ring = nullptr;
        T*                                      base = nullptr;
        std::size_t                             cap  = 0; // elements
        std::atomic<std::size_t>                wr{0};    // monotonically increasing element counts
        std::mutex                              m;
        std::vector<std::shared_ptr<std::atomic<std::size_t>>> readers;
        ~State() { if (ring) gr4hip_ring_destroy(ring); }
        std::size_t min_read() {
            std::lock_guard lk(m);
            std::size_t     r = wr.load(std::memory_order_acquire);
            for (auto& c : readers) r = std::min(r, c->load(std::memory_order_acquire));
            return r;
        }
    };
    std::shared_ptr<State> _s;

    explicit CircularBuffer(std::shared_ptr<State> s) : _s(std::move(s)) {} // another handle on the same ring (Reader / Writer::buffer())

public:
    explicit CircularBuffer(std::size_t min_elements) : _s(std::make_shared<State>()) {
        check(gr4hip_ring_create(&_s->ring, min_elements * sizeof(T)), "gr4hip_ring_create");
        void*       b = nullptr;
        std::size_t bytes = 0;
        check(gr4hip_ring_base(_s->ring, &b), "gr4hip_ring_base");
        check(gr4hip_ring_size(_s->ring, &bytes), "gr4hip_ring_size");
        _s->base = static_cast<T*>(b);
        _s->cap  = bytes / sizeof(T);
    }
    [[nodiscard]] std::size_t size() const noexcept { return _s->cap; }

    class Writer {
        std::shared_ptr<State> _s;
        std::size_t            _reserved = 0, _published = 0;
        friend class CircularBuffer;
        explicit Writer(std::shared_ptr<State> s) : _s(std::move(s)) {}

    public:
        [[nodiscard]] CircularBuffer buffer() const { return CircularBuffer(_s); }                       // BufferWriterLike (Buffer.hpp:88-95)
        [[nodiscard]] std::size_t    nRequestedSamplesToPublish() const noexcept { return _published; } // what the last publish() handed to the readers
        [[nodiscard]] std::size_t available() const { return _s->cap - (_s->wr.load(std::memory_order_relaxed) - _s->min_read()); }
        // device span of n elements at the write cursor, contiguous even across the physical end; empty span if the readers are behind
        [[nodiscard]] std::span<T> tryReserve(std::size_t n) {
            if (n > available()) return {};
            _reserved = n;
            return {_s->base + _s->wr.load(std::memory_order_relaxed) % _s->cap, n};
        }
        [[nodiscard]] std::span<T> reserve(std::size_t n) {
            auto sp = tryReserve(n);
            if (sp.size() != n) throw std::runtime_error("gr::hip::CircularBuffer: reserve exceeds free space");
            return sp;
        }
        void publish(std::size_t n) {
            if (n > _reserved) throw std::runtime_error("gr::hip::CircularBuffer: publish exceeds reservation");
            _reserved  = 0;
            _published = n;
            _s->wr.fetch_add(n, std::memory_order_release);
        }
    };
    class Reader {
        std::shared_ptr<State>                    _s;
        std::shared_ptr<std::atomic<std::size_t>> _rd;
        mutable bool                              _consume_requested = false; // since the last get()
        std::size_t                               _consumed = 0;              // by the last consume()
        friend class CircularBuffer;
        Reader(std::shared_ptr<State> s, std::shared_ptr<std::atomic<std::size_t>> rd) : _s(std::move(s)), _rd(std::move(rd)) {}

    public:
        [[nodiscard]] CircularBuffer buffer() const { return CircularBuffer(_s); }                                    // BufferReaderLike (Buffer.hpp:78-86)
        [[nodiscard]] std::size_t    position() const noexcept { return _rd->load(std::memory_order_relaxed); }       // absolute read cursor (elements)
        [[nodiscard]] std::size_t    nSamplesConsumed() const noexcept { return _consumed; }
        [[nodiscard]] bool           isConsumeRequested() const noexcept { return _consume_requested; }
        [[nodiscard]] std::size_t available() const { return _s->wr.load(std::memory_order_acquire) - _rd->load(std::memory_order_relaxed); }
        [[nodiscard]] std::span<const T> get(std::size_t n) const {
            n                  = std::min(n, available());
            _consume_requested = false;
            return {_s->base + _rd->load(std::memory_order_relaxed) % _s->cap, n};
        }
        [[nodiscard]] bool consume(std::size_t n) {
            if (n > available()) return false;
            _rd->fetch_add(n, std::memory_order_release);
            _consumed          = n;
            _consume_requested = true;
            return true;
        }
    };
    [[nodiscard]] Writer new_writer() { return Writer(_s); }
    [[nodiscard]] Reader new_reader() { // a new reader starts at the current write position (CircularBuffer.hpp semantics)
        auto            rd = std::make_shared<std::atomic<std::size_t>>(_s->wr.load(std::memory_order_acquire));
        std::lock_guard lk(_s->m);
        _s->readers.push_back(rd);
        return Reader(_s, rd);
    }
};

// ---------------------------------------------------------------------------------------------- GPU-domain ports: PortIn<T, GPU> / PortOut<T, GPU>
// The edge between two GPU-domain ports is a CircularBuffer<T> in HBM behind the EdgeBuffer interface the work loop uses: the spans it
// hands out hold DEVICE pointers, contiguous across the wrap.  Host code must not dereference them; blocks with GPU ports pass them to
// kernels.  Crossing between the domains takes an explicit converter block, H2D<T> / D2H<T> below (core/README.md:87-110): the cost of the
// transfer is a visible node of the graph, and everything between the two converters stays in HBM.
// Fan-out (round 5): the reference's edges are one writer -> N readers on ONE buffer (CircularBuffer.hpp:880-946, Graph.hpp:595-690).  A second connection from a
// GPU-domain output port gets a VIEW of the same ring -- its own read cursor (new_reader()), no storage, no copy: a device graph with a tee (spectrum + recorder behind
// one filter) reads the filtered samples twice from the same HBM pages, and the writer's free space is what the slowest reader leaves (the ring's min over its cursors).
// The CPU-domain edges of this layer tee by copying into a mirror buffer (core.hpp); here that would be a device-to-device copy per published span.
template <typename T>
struct DeviceEdgeBuffer final : EdgeBufferBase {
    CircularBuffer<T>                  ring;
    typename CircularBuffer<T>::Writer w;
    typename CircularBuffer<T>::Reader r;
    bool                               is_view = false;                 // a further reader of another edge's ring: reads only
    std::vector<std::shared_ptr<DeviceEdgeBuffer<T>>> views;            // the further readers of THIS edge's ring
    explicit DeviceEdgeBuffer(std::size_t min_elements) : ring(min_elements), w(ring.new_writer()), r(ring.new_reader()) {}
    struct view_of {};
    DeviceEdgeBuffer(DeviceEdgeBuffer& primary, view_of) : ring(primary.w.buffer()), w(ring.new_writer()), r(ring.new_reader()), is_view(true) {
        upstream  = &primary;
        read_pos  = primary.write_pos; // (a new reader starts at the write position: CircularBuffer.hpp semantics)
        write_pos = primary.write_pos;
    }
    [[nodiscard]] std::shared_ptr<DeviceEdgeBuffer<T>> add_reader() {
        auto v = std::make_shared<DeviceEdgeBuffer<T>>(*this, view_of{});
        views.push_back(v);
        mirror_bases.push_back(v); // (tags published on this edge reach every reader's side channel)
        return v;
    }
    [[nodiscard]] std::size_t        available() const noexcept { return r.available(); }
    [[nodiscard]] std::size_t        free_space() const noexcept { return is_view ? 0 : w.available(); }
    [[nodiscard]] std::span<const T> read_span(std::size_t n) const { return r.get(n); }
    [[nodiscard]] std::span<T>       write_span(std::size_t n) {
        if (is_view) throw std::logic_error("DeviceEdgeBuffer: a reader's view of a ring cannot be written");
        return w.reserve(n);
    }
    void                             publish(std::size_t n) {
        if (is_view) throw std::logic_error("DeviceEdgeBuffer: a reader's view of a ring cannot be written");
        w.publish(n);
        advanceWrite(n);
        for (auto& v : views) v->advanceWrite(n);
    }
    void                             consume(std::size_t n) { (void)r.consume(n); advanceRead(n); }
    [[nodiscard]] std::size_t elem_bytes() const noexcept override { return sizeof(T); }
    [[nodiscard]] std::size_t available_items() const noexcept override { return available(); }
    [[nodiscard]] std::size_t free_items() const noexcept override { return free_space(); }
    void read_items(void* dst, std::size_t n) override { // type-erased IO of a DeviceRun at this edge: device -> (pinned) host
        check(gr4hip_memcpy_d2h(dst, read_span(n).data(), n * sizeof(T), nullptr), "d2h");
        check(gr4hip_stream_synchronize(nullptr), "sync");
        consume(n);
    }
    void write_items(const void* src, std::size_t n) override {
        check(gr4hip_memcpy_h2d(write_span(n).data(), src, n * sizeof(T), nullptr), "h2d");
        check(gr4hip_stream_synchronize(nullptr), "sync");
        publish(n);
    }
};

// host -> device converter: CPU-domain input port, GPU-domain output port
template <typename T>
struct H2D : Block<H2D<T>> {
    PortIn<T>       in;
    PortOut<T, GPU> out;
    Size_t          device = 0;
    GR_MAKE_REFLECTABLE(H2D, in, out, device);
    DevBuf          _staging{true};
    std::size_t     _bytes = 0, _staged_bytes = 0; // moved in total / of which through the staging buffer
    work::Status processBulk(std::span<const T> host, std::span<T> dev) {
        try {
            check(gr4hip_set_device(static_cast<int>(device)), "gr4hip_set_device");
            const std::size_t bytes = host.size() * sizeof(T);
            const void*       src   = host.data();
            if (in.buffer->resource() != pinned_resource()) { // pageable edge: one staging copy; an edge from the "hip" provider is DMA-able in place
                std::memcpy(_staging.ensure(bytes), host.data(), bytes);
                src = _staging.p;
                _staged_bytes += bytes;
            }
            check(gr4hip_memcpy_h2d(dev.data(), src, bytes, nullptr), "h2d");
            check(gr4hip_stream_synchronize(nullptr), "sync"); // the span is published only once the copy has landed
            _bytes += bytes;
            return work::Status::OK;
        } catch (const std::exception& e) {
            this->_log(std::string("H2D failed: ") + e.what());
            return work::Status::ERROR;
        }
    }
};
// device -> host converter
template <typename T>
struct D2H : Block<D2H<T>> {
    PortIn<T, GPU> in;
    PortOut<T>     out;
    Size_t         device = 0;
    GR_MAKE_REFLECTABLE(D2H, in, out, device);
    DevBuf         _staging{true};
    work::Status processBulk(std::span<const T> dev, std::span<T> host) {
        try {
            check(gr4hip_set_device(static_cast<int>(device)), "gr4hip_set_device");
            const std::size_t bytes  = dev.size() * sizeof(T);
            const bool        direct = out.connected() && out.buffer->resource() == pinned_resource();
            void*             dst    = direct ? static_cast<void*>(host.data()) : _staging.ensure(bytes);
            check(gr4hip_memcpy_d2h(dst, dev.data(), bytes, nullptr), "d2h");
            check(gr4hip_stream_synchronize(nullptr), "sync");
            if (!direct) std::memcpy(host.data(), dst, bytes);
            return work::Status::OK;
        } catch (const std::exception& e) {
            this->_log(std::string("D2H failed: ") + e.what());
            return work::Status::ERROR;
        }
    }
};

// A hot-path block with GPU-domain ports: OnDevice<fir_filter<float>> has the settings of fir_filter<float> and the ports
// PortIn<T, GPU> / PortOut<U, GPU>; its work() enqueues the block's kernel on device spans -- no copies, no staging.
template <typename BlockT>
requires requires(BlockT& b) { Kernel<BlockT>::make_stage(b); }
struct OnDevice : Block<OnDevice<BlockT>, Resampling<1, 1, false>> {
    using TIn  = typename std::decay_t<decltype(std::declval<BlockT&>().in)>::value_type;
    using TOut = typename std::decay_t<decltype(std::declval<BlockT&>().out)>::value_type;
    PortIn<TIn, GPU>   in;
    PortOut<TOut, GPU> out;
    Size_t             device = 0;
    GR_MAKE_REFLECTABLE(OnDevice, in, out, device);
    BlockT                 block{};  // carries the settings (and is what the stage is built from)
    std::unique_ptr<Stage> _stage;
    gr4hip_stream_t        _stream = nullptr;
    std::size_t            _launches = 0;

    ~OnDevice() {
        _stage.reset();
        if (_stream) gr4hip_stream_destroy(_stream);
    }
    void applySettings(const property_map& settings) { // everything but the wrapper's own keys goes to the wrapped block
        property_map inner;
        for (const auto& [k, v] : settings) {
            if (k == "name") { gr::detail::assign_from(this->name, v); block.name = this->name; }
            else if (k == "device") gr::detail::assign_from(device, v);
            else inner.emplace(k, v);
        }
        block.applySettings(inner);
        _stage.reset(); // rebuilt from the new settings on the next work()
        this->input_chunk_size  = block.input_chunk_size;
        this->output_chunk_size = block.output_chunk_size;
    }
    work::Status processBulk(std::span<const TIn> dev_in, std::span<TOut> dev_out) {
        try {
            check(gr4hip_set_device(static_cast<int>(device)), "gr4hip_set_device");
            if (!_stream) check(gr4hip_stream_create(&_stream), "gr4hip_stream_create");
            if (!_stage) _stage = Kernel<BlockT>::make_stage(block);
            std::size_t produced = 0;
            check(_stage->enqueue(dev_in.data(), dev_in.size(), dev_out.data(), &produced, _stream), "kernel");
            if (produced != dev_out.size()) throw std::runtime_error("device stage produced an unexpected number of samples");
            check(gr4hip_stream_synchronize(_stream), "sync"); // cursors move only after completion (Block.hpp:1989-2026 ordering)
            ++_launches;
            return work::Status::OK;
        } catch (const std::exception& e) {
            this->_log(std::string("device block '") + this->name + "' failed: " + e.what());
            return work::Status::ERROR;
        }
    }
};

// ---------------------------------------------------------------------------------------------- HIP-stream scheduler with chain fusion
// A device run = consecutive device blocks wired 1:1, executed as one unit.
class DeviceRun final : public BlockModel {
    std::vector<std::unique_ptr<Stage>> _stages;
    std::shared_ptr<EdgeBufferBase>     _in_edge, _out_edge;
    std::function<std::size_t()>                                 _avail, _space;
    std::function<void(void*, std::size_t)>                      _read;  // copy n input elements to pinned memory + consume
    std::function<void(const void*, std::size_t)>                _write; // publish n output elements from pinned memory
    gr4hip_ring_t*  _ring = nullptr;
    void*           _ring_base = nullptr;
    std::size_t     _ring_bytes = 0, _ring_wr = 0;
    // three HIP streams, one per pipeline step (ingest copy, kernels, result copy), chained by events: the copy-in of chunk c + 1 and the copy-out of
    // chunk c - 1 run beside the kernels of chunk c.  work() only queues; a chunk's output is published -- and the cursors move -- when its last
    // event has fired (Block.hpp:1989-2026: publish after the work is done), in queue order.
    gr4hip_stream_t _s_in = nullptr, _s_k = nullptr, _s_out = nullptr;
    // a batching run alternates its pieces between two copy streams per direction: a copy of a few hundred KB costs the engine ~10 us of fixed time whatever its size
    // (measured: 256 KiB pieces back to back on ONE stream 14.2 us each = 18.6 GB/s, tools/dbg/api_costs.py) -- two in flight hide it
    gr4hip_stream_t _s_in2 = nullptr, _s_out2 = nullptr;
    std::size_t     _piece_no = 0, _opiece_no = 0;
    static constexpr std::size_t kDepth = 3;
    struct Slot {
        DevBuf         h_in{true}, h_out{true}, d_out;
        gr4hip_event_t in_done = nullptr, k_done = nullptr, out_done = nullptr, in_done2 = nullptr, out_done2 = nullptr;
        std::size_t    n_out = 0;
        std::size_t    seq = 0;               // position of the chunk in the stream of chunks (the lent pieces name their chunk by it)
        void*          direct = nullptr;      // the result copy lands in the output edge's own (page-locked) storage: published in place
        bool           piecewise = false;     // a batching run's chunk: its result leaves in pieces (retire_pieces)
        const void*    d_res = nullptr;       // ... from here (the last stage's output), out_off items of it published so far,
        std::size_t    out_off = 0, q_off = 0;   // q_off: items queued for their copy so far (published + on their way)
        std::size_t    piece_n[2] = {0, 0};      // up to two pieces on their way into the output edge's storage, oldest first ...
        int            piece_s[2] = {0, 0}, pieces = 0; // ... and the result stream each is on
        bool           staged = false, copied = false; // the whole result is (on its way) in h_out; its copy has landed
        bool           busy = false;
        bool           launched = false;      // false: the chunk's samples are on their way into the ring (or there), its kernels are not queued yet
        const void*    d_src = nullptr;       // where the chunk sits in the ring
        std::size_t    n_in = 0;
        property_map   fwd; // tags to publish at the first output sample of this chunk
    };
    std::array<Slot, kDepth> _slots;
    struct Piece { gr4hip_event_t ev; std::size_t n; std::size_t seq; gr4hip_stream_t st = nullptr; }; // an input span the copy engine is reading in the edge's own storage: it has landed when ev has fired -- or (a batching run: no event, a record costs 2.7 us) when its copy stream is idle
    std::deque<Piece>           _lent;
    std::vector<gr4hip_event_t> _ev_free, _ev_all;
    std::size_t                 _next_seq = 0, _launched_seq = 0; // chunks numbered in queue order; every chunk below _launched_seq has its kernels queued
    // batching: an input edge that holds less than a launch is worth -- the reference's default 65 536 items (Graph.hpp:102) -- is drained piece by piece
    // into the device ring and the kernels are launched over ~2^20 items at a time; the result leaves piece by piece as the output edge takes it
    bool                        _batching = false;
    std::size_t                 _batch_items = std::size_t(1) << 21;
    std::size_t     _inplace_chunks = 0, _direct_chunks = 0; // chunks copied straight out of a page-locked input edge / straight into a page-locked output edge
    std::size_t     _q_head = 0, _q_count = 0, _pending_out = 0, _overlapped = 0; // FIFO of busy slots; output items not yet published; chunks queued while another was in flight
    DevBuf          _d_a, _d_b;
    std::string     _name = "device_run";
    ComputeDomain   _domain;
    std::size_t     _in_bytes, _out_bytes, _in_chunk = 1, _out_per_chunk = 1; // smallest input count every stage sees as whole chunks, and what it becomes
    std::size_t     _launches = 0;
    std::string     _desc;
    // the blocks this run stands for (kept alive by Graph::retired) and the stage that realises each: tags address their settings
    struct Member { BlockModel* block; std::size_t stage; };
    std::vector<Member>                                  _members;
    // stage i follows its members' CURRENT settings: `dirty` marks the members that changed; the callee either updates `live` in place (a filter whose absorbed
    // neighbours changed keeps its history) and returns null, or returns a fresh stage
    using Rebuild = std::function<std::unique_ptr<Stage>(std::size_t stage, const std::vector<bool>& dirty, Stage* live)>;
    Rebuild                                              _rebuild;
    std::size_t                                          _tags_forwarded = 0, _stages_rebuilt = 0;

public:
    void set_members(std::vector<Member> members, Rebuild rebuild) { _members = std::move(members); _rebuild = std::move(rebuild); }
    [[nodiscard]] std::size_t tags_forwarded() const { return _tags_forwarded; }
    [[nodiscard]] std::size_t stages_rebuilt() const { return _stages_rebuilt; }
    using RunMember = Member;
    // the edges at both ends are used through their type-erased element IO: a run does not need to know the sample types
    DeviceRun(std::vector<std::unique_ptr<Stage>> stages, std::shared_ptr<EdgeBufferBase> in, std::shared_ptr<EdgeBufferBase> out, ComputeDomain d)
        : _stages(std::move(stages)), _in_edge(in), _out_edge(out), _domain(std::move(d)), _in_bytes(in->elem_bytes()), _out_bytes(out->elem_bytes()) {
        _avail = [in] { return in->available_items(); };
        _space = [out] { return out->free_items(); };
        _read  = [in](void* dst, std::size_t n) { in->read_items(dst, n); };
        _write = [out](const void* src, std::size_t n) { out->write_items(src, n); };
        for (auto& s : _stages) _desc += std::string(_desc.empty() ? "" : " -> ") + std::string(s->kind());
        recompute_rates();
        check(gr4hip_set_device(_domain.index), "gr4hip_set_device");
        for (gr4hip_stream_t* st : {&_s_in, &_s_k, &_s_out, &_s_in2, &_s_out2}) check(gr4hip_stream_create(st), "gr4hip_stream_create");
        for (auto& sl : _slots)
            for (gr4hip_event_t* ev : {&sl.in_done, &sl.k_done, &sl.out_done, &sl.in_done2, &sl.out_done2}) check(gr4hip_event_create(ev), "gr4hip_event_create");
        // GPU-resident double-mapped input ring: kDepth + 1 chunks.  A chunk is a quarter of what the input edge holds (so that the source refills the edge while
        // chunks are in flight), between 2^21 and 2^23 items: fewer, larger copies and launches per sample
        const std::size_t chunk_items = in->memory() == pinned_resource() ? std::clamp<std::size_t>(in->capacity_items() / 4, std::size_t(1) << 21, std::size_t(1) << 23) : std::size_t(1) << 21; // (edges that are staged by host copies overlap better in small chunks)
        check(gr4hip_ring_create(&_ring, (kDepth + 1) * chunk_items * _in_bytes), "gr4hip_ring_create");
        check(gr4hip_ring_base(_ring, &_ring_base), "ring base");
        check(gr4hip_ring_size(_ring, &_ring_bytes), "ring size");
        _batch_items = std::min(_batch_items, chunk_items);
        _batching    = in->capacity_items() > 0 && in->capacity_items() <= _batch_items && options().batch_small_edges;
    }
    ~DeviceRun() override {
        for (gr4hip_stream_t st : {_s_in, _s_in2, _s_k, _s_out, _s_out2})
            if (st) gr4hip_stream_synchronize(st); // nothing may be in flight when the stages and buffers go
        _stages.clear();
        for (auto& sl : _slots)
            for (gr4hip_event_t ev : {sl.in_done, sl.k_done, sl.out_done, sl.in_done2, sl.out_done2})
                if (ev) gr4hip_event_destroy(ev);
        for (gr4hip_event_t ev : _ev_all) gr4hip_event_destroy(ev);
        if (_ring) gr4hip_ring_destroy(_ring);
        for (gr4hip_stream_t st : {_s_in, _s_in2, _s_k, _s_out, _s_out2})
            if (st) gr4hip_stream_destroy(st);
    }
    [[nodiscard]] std::size_t overlapped_chunks() const { return _overlapped; }
    [[nodiscard]] std::size_t inplace_chunks() const { return _inplace_chunks; }
    [[nodiscard]] std::size_t direct_chunks() const { return _direct_chunks; }
    // ---- input spans lent to the copy engine (pieces, oldest first).  A piece goes back to the edge when its copy has landed and -- edges that hold whole chunks --
    // its chunk's kernels are queued (a chunk whose launch fails goes back to the edge whole); a batching run (below) gives it back as soon as the copy has landed:
    // the edge is smaller than the batch, the samples wait for their launch in the device ring
    [[nodiscard]] bool releasable(const Piece& pc) const { return _batching || pc.seq < _launched_seq; }
    bool piece_landed(const Piece& pc, bool wait) {
        if (wait) { check(pc.ev ? gr4hip_event_synchronize(pc.ev) : gr4hip_stream_synchronize(pc.st), "sync"); return true; }
        int done = 0;
        check(pc.ev ? gr4hip_event_query(pc.ev, &done) : gr4hip_stream_query(pc.st, &done), "query");
        return done != 0;
    }
    void piece_gone(const Piece& pc) { if (pc.ev) _ev_free.push_back(pc.ev); }
    // the oldest piece still lent: wait for its copy and give it back; returns the items released (0: nothing that may go back is lent)
    std::size_t release_oldest_input() {
        if (_lent.empty() || !releasable(_lent.front())) return 0;
        const Piece pc = _lent.front();
        piece_landed(pc, true);
        _in_edge->consume_items(pc.n);
        piece_gone(pc);
        _lent.pop_front();
        return pc.n;
    }
    // lent pieces go back to the edge, oldest first, as their copies land (wait: block until all have)
    std::size_t release_inputs(bool wait) { // returns the items that went back
        std::size_t freed = 0;
        if (wait) launch_pending(0); // (everything lent is to go back: the chunks that hold spans run first)
        while (!_lent.empty() && releasable(_lent.front())) {
            const Piece pc = _lent.front();
            if (!piece_landed(pc, wait)) return freed;
            _in_edge->consume_items(pc.n);
            piece_gone(pc);
            _lent.pop_front();
            freed += pc.n;
        }
        return freed;
    }
    gr4hip_event_t piece_event() {
        if (_ev_free.empty()) { gr4hip_event_t ev = nullptr; check(gr4hip_event_create(&ev), "gr4hip_event_create"); _ev_all.push_back(ev); return ev; }
        gr4hip_event_t ev = _ev_free.back();
        _ev_free.pop_back();
        return ev;
    }
    // rate bookkeeping (Resampling<>, Block.hpp:1576-1636, across the whole run): walking back from the last stage, `need` is the count a
    // stage's output must be a multiple of; it produces out_chunk per in_chunk
    void recompute_rates() {
        std::size_t need = 1;
        for (auto it = _stages.rbegin(); it != _stages.rend(); ++it) {
            const std::size_t k = need / std::gcd(need, (*it)->out_chunk); // chunks so that k * out_chunk is a multiple of need
            need                = k * (*it)->in_chunk;
        }
        _in_chunk      = need;
        _out_per_chunk = out_count(_in_chunk);
    }
    [[nodiscard]] std::size_t out_count(std::size_t n_in) const { // elements leaving the last stage for n_in entering the first
        for (auto& s : _stages) n_in = n_in / s->in_chunk * s->out_chunk;
        return n_in;
    }
    std::string_view           description() const { return _desc; }
    [[nodiscard]] std::size_t  launches() const { return _launches; }
    const std::vector<std::unique_ptr<Stage>>& stages() const { return _stages; }

    // the kernels and the result copy of a chunk whose samples are on their way into the ring.  Ingest and launch are separate steps: work() queues the copy of
    // chunk c + 1 BEFORE it launches chunk c, so that the link stays busy while a stage's enqueue is busy.  A batching run's result goes out piece by piece (retire):
    // only the wait for the kernels is queued on the result stream here
    void launch(Slot& sl) {
        if (sl.piecewise) { // a batch: its pieces went over both copy streams; the events go behind the newest piece of each (pieces of the NEXT batch may sit in front of them: small)
            check(gr4hip_event_record(sl.in_done, _s_in), "event record");
            check(gr4hip_event_record(sl.in_done2, _s_in2), "event record");
            check(gr4hip_stream_wait_event(_s_k, sl.in_done2), "stream wait");
        }
        check(gr4hip_stream_wait_event(_s_k, sl.in_done), "stream wait");
        const void* cur = sl.d_src;
        std::size_t cnt = sl.n_in;
        for (std::size_t i = 0; i < _stages.size(); ++i) { // stages run back-to-back on the kernel stream; intermediates stay in HBM
            const std::size_t expect = cnt / _stages[i]->in_chunk * _stages[i]->out_chunk;
            DevBuf&           dst    = i + 1 == _stages.size() ? sl.d_out : ((i % 2) ? _d_b : _d_a); // the last stage writes the chunk's own result buffer
            std::size_t       out    = 0;
            check(_stages[i]->enqueue(cur, cnt, dst.ensure(std::max<std::size_t>(expect, 1) * _stages[i]->out_bytes), &out, _s_k), "stage");
            if (out != expect) throw std::runtime_error("stage '" + std::string(_stages[i]->kind()) + "' produced an unexpected number of samples");
            cur = dst.p;
            cnt = out;
            ++_launches;
        }
        if (cnt != sl.n_out) throw std::runtime_error("device run: a chunk produced an unexpected number of samples");
        check(gr4hip_event_record(sl.k_done, _s_k), "event record");
        check(gr4hip_stream_wait_event(_s_out, sl.k_done), "stream wait");
        if (sl.piecewise) check(gr4hip_stream_wait_event(_s_out2, sl.k_done), "stream wait");
        sl.d_res = cur;
        if (!sl.piecewise || _out_edge->memory() != pinned_resource()) { // the whole result in one copy: into the reserved span of a page-locked edge, or into the slot's own page-locked buffer
            check(gr4hip_memcpy_d2h(sl.direct ? sl.direct : sl.h_out.ensure(cnt * _out_bytes), cur, cnt * _out_bytes, _s_out), "d2h");
            check(gr4hip_event_record(sl.out_done, _s_out), "event record");
            sl.staged = !sl.direct;
        }
        sl.launched   = true;
        _launched_seq = sl.seq + 1;
    }
    // launches every ingested chunk except the newest `keep` ones, oldest first
    void launch_pending(std::size_t keep) {
        for (std::size_t i = 0; i + keep < _q_count; ++i) {
            Slot& sl = _slots[(_q_head + i) % kDepth];
            if (!sl.launched) launch(sl);
        }
    }
    void finish_slot(Slot& sl) {
        sl.direct = nullptr;
        sl.busy = sl.launched = sl.staged = false;
        sl.out_off = sl.q_off = 0;
        sl.pieces = 0;
        sl.fwd.clear();
        _q_head = (_q_head + 1) % kDepth;
        --_q_count;
    }
    // a batching run's oldest chunk: its result leaves in pieces of at most half the output edge -- straight into the edge's page-locked storage where there is room (the sink
    // drains one half while the copy engine fills the other), or out of the slot's staging buffer into an ordinary edge.  Publishes what has landed, queues the next piece, never
    // waits for room in the edge (the sink runs on this thread).  only_if_done: never waits for a copy either.
    std::size_t retire_pieces(Slot& sl, bool only_if_done) {
        std::size_t       pub  = 0;
        const bool        pin  = _out_edge->memory() == pinned_resource();
        static const int opiece_div = [] { const char* e = std::getenv("GR4HIP_RUN_OPIECE_DIV"); return e ? std::max(1, std::atoi(e)) : 1; }(); // developer knob (2: half the edge per piece -- measured slower at every edge size: the copy engine spends ~10 us per operation whatever its size, one operation at a time per direction)
        const std::size_t half = std::max<std::size_t>(_out_edge->capacity_items() / opiece_div, 1);
        const auto landed = [&](gr4hip_event_t ev) {
            if (!only_if_done) { check(gr4hip_event_synchronize(ev), "event sync"); return true; }
            int done = 0;
            check(gr4hip_event_query(ev, &done), "event query");
            return done != 0;
        };
        const auto idle = [&](gr4hip_stream_t st) { // a piece has landed when the result stream it is on has nothing left to do (one piece per stream in flight)
            if (!only_if_done) { check(gr4hip_stream_synchronize(st), "stream sync"); return true; }
            int done = 0;
            check(gr4hip_stream_query(st, &done), "stream query");
            return done != 0;
        };
        for (;;) {
            if (sl.staged && !sl.copied) { // the whole result on its way into the slot's staging buffer
                if (!landed(sl.out_done)) return pub;
                sl.copied = true;
            }
            while (sl.pieces) { // pieces that have landed appear on the edge, in order
                if (!idle(sl.piece_s[0] ? _s_out2 : _s_out)) break;
                if (sl.out_off == 0 && !sl.fwd.empty()) { _out_edge->publishTag(sl.fwd, 0); ++_tags_forwarded; }
                _out_edge->publish_reserved(sl.piece_n[0]);
                sl.out_off += sl.piece_n[0];
                pub += sl.piece_n[0];
                sl.piece_n[0] = sl.piece_n[1]; sl.piece_s[0] = sl.piece_s[1];
                --sl.pieces;
            }
            if (sl.out_off == sl.n_out) { finish_slot(sl); return pub; }
            bool queued = false;
            if (pin) {
                while (sl.pieces < 2 && sl.q_off < sl.n_out) {
                    if (sl.pieces == 1 && sl.piece_s[0] == (int)(_opiece_no & 1)) ++_opiece_no; // (the stream the piece in flight is NOT on: one piece per stream)
                    const std::size_t n = std::min({sl.n_out - sl.q_off, half, _out_edge->free_items()});
                    void* dst = n ? _out_edge->reserve_items(n) : nullptr;
                    if (!dst) break; // the edge is full: the sink's turn
                    const int which = (int)(_opiece_no++ & 1);
                    check(gr4hip_memcpy_d2h(dst, static_cast<const char*>(sl.d_res) + sl.q_off * _out_bytes, n * _out_bytes, which ? _s_out2 : _s_out), "d2h");
                    sl.piece_n[sl.pieces] = n; sl.piece_s[sl.pieces] = which;
                    ++sl.pieces;
                    sl.q_off += n;
                    ++_direct_chunks;
                    queued = true;
                }
            } else {
                const std::size_t n = std::min(sl.n_out - sl.out_off, _out_edge->free_items());
                if (n) {
                    if (sl.out_off == 0 && !sl.fwd.empty()) { _out_edge->publishTag(sl.fwd, 0); ++_tags_forwarded; }
                    _write(static_cast<const char*>(sl.h_out.p) + sl.out_off * _out_bytes, n);
                    sl.out_off += n; sl.q_off = sl.out_off;
                    pub += n;
                    if (sl.out_off == sl.n_out) { finish_slot(sl); return pub; }
                }
            }
            if (only_if_done || (!sl.pieces && !queued)) return pub; // nothing more can happen without the sink (or without waiting)
        }
    }

    // publish the oldest queued chunk (blocking until its result copy has landed unless only_if_done); returns the items published
    std::size_t retire(bool only_if_done) {
        if (_q_count == 0) return 0;
        Slot& sl = _slots[_q_head];
        if (!sl.launched) {
            if (only_if_done) return 0;
            launch(sl);
        }
        if (sl.piecewise) return retire_pieces(sl, only_if_done);
        if (only_if_done) {
            int done = 0;
            check(gr4hip_event_query(sl.out_done, &done), "event query");
            if (!done) return 0;
        } else {
            check(gr4hip_event_synchronize(sl.out_done), "event sync");
        }
        while (!_lent.empty() && _lent.front().seq <= sl.seq) { // the result has landed, so has the input (slots retire oldest first)
            _in_edge->consume_items(_lent.front().n);
            piece_gone(_lent.front());
            _lent.pop_front();
        }
        if (!sl.fwd.empty()) { _out_edge->publishTag(sl.fwd, 0); ++_tags_forwarded; }
        const std::size_t n = sl.n_out;
        if (sl.direct) _out_edge->publish_reserved(n);
        else {
            void* dst = n * _out_bytes >= (std::size_t(2) << 20) ? _out_edge->reserve_items(n) : nullptr; // large chunk into a pageable edge: the copy threads share it
            if (dst) { CopyPool::instance().copy(dst, sl.h_out.p, n * _out_bytes); _out_edge->publish_reserved(n); }
            else _write(sl.h_out.p, n);
            _pending_out -= n;
        }
        finish_slot(sl);
        return n;
    }
    // everything queued leaves (a stage is about to be replaced, tags are about to be read relative to the read position).  false: the output edge is full and the
    // sink has to run first -- only a batching run can be stuck like that (a chunk-per-slot run never queues more than the output edge takes)
    bool drain(std::size_t& published) {
        while (_q_count) {
            const std::size_t before = _q_count;
            const std::size_t r      = retire(false);
            published += r;
            if (_q_count == before && r == 0) return false;
        }
        return true;
    }

    work::Result work(std::size_t requested) override {
        std::size_t held_lent = 0, held_reserved = 0; // spans taken from the edges for a chunk that is not queued yet (given back if the launch fails)
        try {
            check(gr4hip_set_device(_domain.index), "gr4hip_set_device"); // runs on several devices share the scheduler thread: the current device is per call
            std::size_t published = 0;
            while (const std::size_t r = retire(true)) published += r; // whatever has finished since the last call
            if (_batching && !_in_edge->tags.empty() && _q_count) { // tags around: a batching run goes chunk by chunk (nothing in flight while settings may change)
                launch_pending(0);
                if (!drain(published)) return {requested, published, published ? work::Status::OK : work::Status::INSUFFICIENT_OUTPUT_ITEMS};
            }
            const std::size_t freed_now = release_inputs(!_in_edge->tags.empty()); // tags are addressed relative to the read position: with tags around, every lent span is returned first
            // the launch's tag: the one on its first sample -- launches end where the next tag starts -- or, when a whole input chunk (a frame, a
            // decimation group) has to span tags, all of them merged (Block.hpp:1511-1530).  Settings-by-tag for the member blocks first (only the
            // stages of members that changed are rebuilt, the others keep their state), then forwarded across the whole run like across one block:
            // "gr:" keys, at the first output sample, gr:sample_rate scaled by the run's rate change
            property_map fwd;
            if (!_in_edge->tags.empty()) {
                const auto apply = [&](const property_map& map) {
                    std::vector<bool> dirty(_stages.size(), false), changed(_members.size(), false);
                    for (std::size_t m = 0; m < _members.size(); ++m)
                        if (_members[m].block->apply_tag_settings(map)) dirty[_members[m].stage] = changed[m] = true;
                    if (std::find(dirty.begin(), dirty.end(), true) == dirty.end()) return;
                    while (_q_count) published += retire(false); // a stage is replaced: nothing of the old one may be in flight (a batching run has drained above)
                    for (std::size_t i = 0; i < _stages.size(); ++i)
                        if (dirty[i] && _rebuild) {
                            if (auto fresh = _rebuild(i, changed, _stages[i].get())) _stages[i] = std::move(fresh);
                            ++_stages_rebuilt;
                        }
                    _desc.clear();
                    for (auto& st : _stages) _desc += std::string(_desc.empty() ? "" : " -> ") + std::string(st->kind());
                    recompute_rates(); // the launch below is sized with the new chunking
                };
                property_map merged;
                if (const Tag* t = _in_edge->tagAtReadPosition()) { merged = t->map; apply(merged); }
                if (_in_edge->samplesUntilNextTag() < _in_chunk && _avail() >= _in_chunk) { // tags inside the one chunk this launch cannot be shorter than
                    merged = _in_edge->mergedTags(_in_chunk);
                    apply(merged);
                }
                for (const auto& [key, value] : merged) {
                    if (!std::string_view(key).starts_with(GR_TAG_PREFIX)) continue;
                    const float* rate = tag::settingsKey(key) == tag::SAMPLE_RATE && _out_per_chunk != _in_chunk ? std::get_if<float>(&value) : nullptr;
                    if (rate) fwd.insert_or_assign(key, static_cast<float>(_out_per_chunk) / static_cast<float>(_in_chunk) * *rate);
                    else fwd.insert_or_assign(key, value);
                }
            }
            // ---- a batching run (the input edge holds less than a launch is worth: the reference's default 65 536-item edges, Graph.hpp:102): the piece that has arrived joins
            // the newest chunk while that one is not launched, carries no tag of its own and is short of the batch size -- one launch per ~2^20 items instead of one per edge-full
            const std::size_t batch_items = std::max(_batch_items / _in_chunk, std::size_t(1)) * _in_chunk;
            Slot* open = nullptr;
            if (_batching && _q_count && fwd.empty()) {
                Slot& newest = _slots[(_q_head + _q_count - 1) % kDepth];
                if (!newest.launched && newest.n_in < batch_items) open = &newest;
            }
            if (!open && _q_count == kDepth) {
                published += retire(false); // all slots queued: wait for the oldest
                if (_q_count == kDepth) return {requested, published, published ? work::Status::OK : work::Status::INSUFFICIENT_OUTPUT_ITEMS}; // (batching: the sink has to make room first)
            }
            std::size_t n = std::min({_avail(), requested, _batching ? (open ? batch_items - open->n_in : batch_items) : _ring_bytes / _in_bytes / (kDepth + 1)}); // kDepth chunks in flight never wrap onto each other in the ring
            n = std::min(n, std::max(_in_edge->samplesUntilNextTag(), _in_chunk)); // a launch ends where the next tag starts (Block.hpp:1511-1530): tags sit on launch boundaries
            if (_batching) {
                // half the edge per piece: the source refills one half while the copy engine reads the other
                static const int piece_div = [] { const char* e = std::getenv("GR4HIP_RUN_PIECE_DIV"); return e ? std::max(1, std::atoi(e)) : 1; }(); // developer knob (2: half the edge per piece -- measured slower at every edge size: the copy engine spends ~10 us per operation whatever its size, one operation at a time per direction)
                if (_in_edge->capacity_items() / piece_div >= _in_chunk) n = std::min(n, _in_edge->capacity_items() / piece_div);
                n = n / _in_chunk * _in_chunk;
            } else {
                const std::size_t space = _space() - std::min(_space(), _pending_out);  // the output edge minus what queued chunks will publish (reserved spans are already off it)
                n = std::min(n / _in_chunk, space / std::max<std::size_t>(1, _out_per_chunk)) * _in_chunk; // whole chunks that also fit the output edge
            }
            if (n == 0) {
                if (_batching) {
                    if (freed_now) return {requested, std::max(published, freed_now), work::Status::OK}; // a piece has just gone back: the source writes into its place while the next one is on the link
                    // nothing new has arrived.  A lent piece is what keeps a small edge full: it goes back as soon as ITS copy has landed and the source runs; with nothing
                    // lent the input has run dry for now -- what has been gathered is launched (a slow source is served chunk by chunk, a fast one in full batches)
                    if (const std::size_t freed = release_oldest_input()) return {requested, std::max(published, freed), work::Status::OK};
                    launch_pending(0);
                    if (_q_count) {
                        const std::size_t r = retire(false);
                        published += r;
                        if (r) return {requested, published, work::Status::OK};
                        if (published) return {requested, published, work::Status::OK};
                        return {requested, 0, _avail() < _in_chunk && !_in_edge->done() && _out_edge->free_items() ? work::Status::INSUFFICIENT_INPUT_ITEMS : work::Status::INSUFFICIENT_OUTPUT_ITEMS};
                    }
                } else {
                    launch_pending(0); // nothing new to copy: whatever has been ingested runs now
                    if (_q_count) { // nothing new to queue
                        // input lent to the copy engine is what keeps a small edge full: give the oldest such span back as soon as ITS copy has landed and let the
                        // source refill the edge while the kernels and the result copy of that chunk still run; otherwise publish the oldest chunk
                        if (const std::size_t freed = release_oldest_input()) return {requested, std::max(published, freed), work::Status::OK}; // (progress: the source can write again)
                        published += retire(false);
                        return {requested, published, work::Status::OK};
                    }
                }
                if (published) return {requested, published, work::Status::OK};
                if (_avail() < _in_chunk && _in_edge->done()) {
                    _out_edge->producer_done = true;
                    return {requested, 0, work::Status::DONE};
                }
                return {requested, 0, _avail() < _in_chunk ? work::Status::INSUFFICIENT_INPUT_ITEMS : work::Status::INSUFFICIENT_OUTPUT_ITEMS};
            }
            // a page-locked output edge ("hip" provider) takes the result copy in its own storage; at the end of the storage the edge must compact
            // first, which it only does with nothing in flight.  (A batching run's results leave piece by piece: retire_pieces)
            void* direct = nullptr;
            if (!_batching && _out_edge->memory() == pinned_resource()) {
                const std::size_t n_res = out_count(n);
                direct = _out_edge->reserve_items(n_res);
                if (!direct && _q_count) {
                    while (_q_count) published += retire(false);
                    direct = _out_edge->reserve_items(n_res);
                }
                if (direct) held_reserved = n_res;
            }
            const bool fresh = open == nullptr;
            Slot& sl = fresh ? _slots[(_q_head + _q_count) % kDepth] : *open;
            if (fresh && _q_count) ++_overlapped;
            if (fresh) sl.seq = _next_seq++;
            // samples land in HBM: pinned staging -> hipMemcpyAsync -> the double-mapped ring (a wrapping span stays contiguous)
            char*       d_in    = static_cast<char*>(_ring_base) + _ring_wr;
            const void* lent = _in_edge->lend_items(n);
            if (lent) held_lent = n;
            if (lent && _in_edge->memory() == pinned_resource()) { // page-locked storage ("hip" provider): the copy engine reads the edge in place;
                gr4hip_stream_t sin = _batching && (_piece_no++ & 1) ? _s_in2 : _s_in;
                check(gr4hip_memcpy_h2d(d_in, lent, n * _in_bytes, sin), "h2d"); // the span goes back to the edge once the copy has landed (release_inputs)
                if (_batching) _lent.push_back({nullptr, n, sl.seq, sin});
                else {
                    gr4hip_event_t ev = piece_event();
                    check(gr4hip_event_record(ev, sin), "event record");
                    _lent.push_back({ev, n, sl.seq});
                }
                ++_inplace_chunks;
            } else {
                // pageable edge: staged through page-locked memory (the copy threads share a large piece).  A chunk-per-slot run keeps the span lent until the chunk is launched,
                // like an in-place one -- a chunk whose launch fails goes back to the edge whole; a batching run hands it back at once (the staging buffer has it)
                const std::size_t off = fresh ? 0 : sl.n_in * _in_bytes;
                char*             stage = static_cast<char*>(sl.h_in.ensure((_batching ? batch_items : n) * _in_bytes)) + off;
                if (lent) {
                    CopyPool::instance().copy(stage, lent, n * _in_bytes);
                    if (_batching) _in_edge->consume_items(n);
                    else {
                        gr4hip_event_t ev = piece_event();
                        check(gr4hip_memcpy_h2d(d_in, stage, n * _in_bytes, _s_in), "h2d");
                        check(gr4hip_event_record(ev, _s_in), "event record");
                        _lent.push_back({ev, n, sl.seq});
                        stage = nullptr;
                    }
                } else {
                    _read(stage, n);
                }
                if (stage) check(gr4hip_memcpy_h2d(d_in, stage, n * _in_bytes, _s_in), "h2d");
            }
            if (!_batching) check(gr4hip_event_record(sl.in_done, _s_in), "event record"); // what the kernel stream waits for (a batch: recorded when it is launched, behind its last piece)
            _ring_wr = (_ring_wr + n * _in_bytes) % _ring_bytes;
            if (fresh) {
                sl.d_src     = d_in;
                sl.n_in      = n;
                sl.busy      = true;
                sl.launched  = false;
                sl.staged = sl.copied = false;
                sl.out_off = sl.q_off = 0;
                sl.pieces    = 0;
                sl.piecewise = _batching;
                sl.direct    = direct;
                sl.fwd       = std::move(fwd);
                ++_q_count;
            } else {
                sl.n_in += n;
            }
            sl.n_out = out_count(sl.n_in);
            if (direct) ++_direct_chunks;
            else if (!_batching) _pending_out += sl.n_out;
            held_lent = held_reserved = 0; // (the slot owns the spans from here on)
            launch_pending(1);             // the chunk before this one: its kernels go out while this chunk's copy is on the link
            if (_batching && sl.n_in >= batch_items) launch_pending(0); // a full batch does not wait for the next piece
            return {requested, n, work::Status::OK};
        } catch (const std::exception& e) {
            std::cerr << "[gr::hip] device run failed: " << e.what() << "\n";
            // the failed chunk's spans go back to their edges untouched (a graph that tolerates ERROR must not find free_items() / available_items() shrunk
            // for good) -- once nothing queued on the streams can still read or write them
            bool unlaunched = false;
            for (std::size_t i = 0; i < _q_count; ++i) unlaunched = unlaunched || !_slots[(_q_head + i) % kDepth].launched;
            if (held_lent || held_reserved || unlaunched) {
                for (gr4hip_stream_t st : {_s_in, _s_in2, _s_k, _s_out, _s_out2}) (void)gr4hip_stream_synchronize(st);
                if (held_lent) {
                    if (!_lent.empty() && _lent.back().n == held_lent && _lent.back().seq + 1 == _next_seq) { piece_gone(_lent.back()); _lent.pop_back(); } // (the piece of the call that failed)
                    _in_edge->unlend_items(held_lent);
                }
                if (held_reserved) _out_edge->unreserve_items(held_reserved);
                // ingested chunks whose launch failed (or never happened) are the newest entries of the queue.  A chunk-per-slot run hands their spans back (newest first) and
                // forgets them; a batching run has given the spans back already -- the samples are in the device ring: the chunk stays queued and is launched again by the next call
                while (!_batching && _q_count && !_slots[(_q_head + _q_count - 1) % kDepth].launched) {
                    Slot& sl = _slots[(_q_head + _q_count - 1) % kDepth];
                    while (!_lent.empty() && _lent.back().seq == sl.seq) { _in_edge->unlend_items(_lent.back().n); piece_gone(_lent.back()); _lent.pop_back(); }
                    if (sl.direct) _out_edge->unreserve_items(sl.n_out);
                    else _pending_out -= std::min(_pending_out, sl.n_out);
                    sl.direct = nullptr;
                    sl.busy   = false;
                    sl.fwd.clear();
                    --_q_count;
                    _ring_wr = static_cast<std::size_t>(static_cast<const char*>(sl.d_src) - static_cast<const char*>(_ring_base)) % _ring_bytes; // (the ring takes the next chunk where this one began)
                }
            }
            return {requested, 0, work::Status::ERROR};
        }
    }
    std::string_view     name() const override { return _name; }
    std::string_view     type_name() const override { return "gr::hip::DeviceRun"; }
    const ComputeDomain& compute_domain() const override { return _domain; }
    void*                raw() override { return this; }
    std::type_index      port_type(std::string_view) override { return typeid(void); }
    std::shared_ptr<EdgeBufferBase> make_edge(std::string_view, std::size_t, std::pmr::memory_resource*) override { return nullptr; }
    bool attach_input(std::string_view, std::shared_ptr<EdgeBufferBase>) override { return false; }
    std::vector<std::shared_ptr<EdgeBufferBase>> input_edges() override { return {_in_edge}; }
    std::vector<std::shared_ptr<EdgeBufferBase>> output_edges() override { return {_out_edge}; }
};

// Builds the stage list of a linear device chain, fusing fir_filter<complex<float>> -> PowerSpectrum into gr4hip_chain.
// Usage (see host/tests): fuse_chain(graph, fir, spectrum) replaces both blocks by one DeviceRun in the graph's block list.
template <typename First, typename... Rest>
DeviceRun& fuse_chain(Graph& g, First& first, Rest&... rest) {
    std::vector<std::unique_ptr<Stage>> stages;
    auto& last = std::get<sizeof...(Rest)>(std::tie(first, rest...));
    {
        std::vector<std::unique_ptr<Stage>> made;
        made.push_back(Kernel<First>::make_stage(first));
        (made.push_back(Kernel<Rest>::make_stage(rest)), ...);
        for (auto& grp : fuse_stages(std::move(made))) stages.push_back(std::move(grp.stage)); // neighbours with a fused kernel share one launch
    }
    if (!first.in.connected() || !last.out.connected()) throw std::invalid_argument("fuse_chain: connect the chain to its neighbours first");
    auto  run = std::make_unique<DeviceRun>(std::move(stages), first.in.buffer, last.out.buffer, ComputeDomain::parse(first.compute_domain));
    auto& ref = *run;
    auto& blocks = g.blocks();
    const void* members[] = {static_cast<const void*>(&first), static_cast<const void*>(&rest)...};
    // the run takes the place of its first member; the members themselves leave the schedule (their state lives in the stages)
    std::size_t first_pos = blocks.size();
    for (std::size_t i = 0; i < blocks.size(); ++i)
        if (blocks[i]->raw() == members[0]) first_pos = i;
    if (first_pos == blocks.size()) throw std::invalid_argument("fuse_chain: block is not part of this graph");
    std::vector<std::unique_ptr<BlockModel>> kept;
    for (std::size_t i = 0; i < blocks.size(); ++i) {
        const bool member = std::find(std::begin(members), std::end(members), blocks[i]->raw()) != std::end(members);
        if (i == first_pos) kept.push_back(std::move(run));
        if (!member) kept.push_back(std::move(blocks[i]));
        else g.retired().push_back(std::move(blocks[i])); // keep the objects alive: callers hold references to them
    }
    blocks = std::move(kept);
    return ref;
}

// ---------------------------------------------------------------------------------------------- fusion planner
// The run-time counterpart of the reference's compile-time Merge<> (BlockMerging.hpp:136-320): every maximal chain of blocks that
//   * ask for this device (compute_domain "gpu:hip[:i]", all the same),  * have a device kernel (BlockModel::make_device_stage), and
//   * are wired 1:1 (one input edge, one output edge, the edge between two members has no other reader)
// is replaced by ONE DeviceRun: samples enter HBM once, the stages run back to back on one stream, and adjacent stages with a fused
// kernel collapse into it (fir_filter<complex<float>> -> PowerSpectrum = gr4hip_chain, any window).  Returns the runs it created.
// run_edge_items: the edges at both ends of every run are grown to at least this many items while they are still empty (same memory resource; 0: left
// alone).  A run moves its input over the link in whole chunks and keeps several in flight; on the reference's default 65536-item edges a chunk IS the edge and
// source, copies and kernels take turns (0.75 Gsamples/s host-fed where 2^22-item edges give 5.7: profiles/r02_host_feed.txt).  Larger page-locked edges that a
// driver fills by DMA are worth having (a run moves a quarter of such an edge per chunk, up to 64 MiB: 4.8 / 6.2 Gsamples/s on 2^22 / 2^24-item edges); edges the
// CPU writes sample by sample are not (cache footprint: 3.4 -> 2.6 with a copying source): profiles/r03_host_feed.txt.
inline std::vector<DeviceRun*> plan(Graph& g, std::size_t min_blocks = 2, std::size_t run_edge_items = std::size_t(1) << 22) {
    auto& blocks = g.blocks();
    const auto eligible = [](BlockModel& b) {
        const auto& d = b.compute_domain();
        return d.is_device() && (d.backend.empty() || d.backend == "hip") && b.input_edges().size() == 1 && b.output_edges().size() == 1 &&
               b.input_edges()[0] && b.output_edges()[0];
    };
    const auto readers = [&](const std::shared_ptr<EdgeBufferBase>& e) {
        std::size_t n = 0;
        for (auto& b : blocks)
            for (auto& in : b->input_edges()) n += in == e;
        return n;
    };
    const auto consumer = [&](const std::shared_ptr<EdgeBufferBase>& e) -> BlockModel* {
        for (auto& b : blocks)
            for (auto& in : b->input_edges())
                if (in == e) return b.get();
        return nullptr;
    };
    const auto producer = [&](const std::shared_ptr<EdgeBufferBase>& e) -> BlockModel* {
        for (auto& b : blocks)
            for (auto& out : b->output_edges())
                if (out == e) return b.get();
        return nullptr;
    };
    std::vector<std::vector<BlockModel*>> chains;
    std::vector<BlockModel*>              taken;
    for (auto& bp : blocks) {
        BlockModel* b = bp.get();
        if (!eligible(*b) || std::find(taken.begin(), taken.end(), b) != taken.end()) continue;
        // walk back to the head of the chain this block belongs to
        const auto joins = [&](BlockModel* up, BlockModel* down) {
            return up && down && eligible(*up) && eligible(*down) && up->output_edges()[0] == down->input_edges()[0] && readers(up->output_edges()[0]) == 1 &&
                   up->output_edges()[0]->mirror_bases.empty() && // a tee'd output has readers outside the chain: its samples must reach the host edge

                   up->compute_domain().index == down->compute_domain().index;
        };
        BlockModel* head = b;
        while (BlockModel* up = producer(head->input_edges()[0]))
            if (joins(up, head)) head = up; else break;
        std::vector<BlockModel*> chain{head};
        while (BlockModel* down = consumer(chain.back()->output_edges()[0]))
            if (joins(chain.back(), down)) chain.push_back(down); else break;
        for (auto* m : chain) taken.push_back(m);
        if (chain.size() >= min_blocks) chains.push_back(std::move(chain));
    }
    // a stage created through the type-erased hook (the deleter travels with the shared_ptr), owned by the run
    struct Holder final : Stage {
        std::shared_ptr<Stage> s;
        explicit Holder(std::shared_ptr<Stage> p) : s(std::move(p)) { sync(); }
        void sync() { in_bytes = s->in_bytes; out_bytes = s->out_bytes; in_chunk = s->in_chunk; out_chunk = s->out_chunk; }
        int              enqueue(const void* i, std::size_t n, void* o, std::size_t* no, gr4hip_stream_t st) override { return s->enqueue(i, n, o, no, st); }
        std::string_view kind() const override { return s->kind(); }
        const EwiseProgram* program() const override { return s->program(); }
        bool absorb(const EwiseProgram& p, bool before) override { return s->absorb(p, before); }
        void clear_absorbed() override { s->clear_absorbed(); }
        Stage* self() override { return s->self(); }
    };
    std::vector<DeviceRun*> runs;
    for (auto& chain : chains) {
        // member stages from the members' CURRENT settings, then fused: per-sample neighbours into one program or into the launch of the filter next to them,
        // fir -> Decimator into the polyphase filter, fir_filter<complex<float>> -> PowerSpectrum into the fused chain kernel (fuse_stages)
        const auto make_members = [chain](std::size_t first, std::size_t count) {
            std::vector<std::unique_ptr<Stage>> made;
            for (std::size_t m = first; m < first + count; ++m) {
                auto st = std::static_pointer_cast<Stage>(chain[m]->make_device_stage());
                if (!st) return std::vector<std::unique_ptr<Stage>>{};
                made.push_back(std::make_unique<Holder>(std::move(st)));
            }
            return made;
        };
        auto made = make_members(0, chain.size());
        if (made.size() != chain.size()) continue; // a member without a device kernel: leave the chain to the per-block seam
        auto groups = fuse_stages(std::move(made));
        struct GroupInfo { std::size_t first, count, anchor; };
        std::vector<GroupInfo>            info;
        std::vector<DeviceRun::RunMember> members;
        std::vector<std::unique_ptr<Stage>> fused;
        for (auto& grp : groups) {
            info.push_back({grp.members.front(), grp.members.size(), grp.anchor});
            for (std::size_t m : grp.members) members.push_back({chain[m], fused.size()});
            fused.push_back(std::move(grp.stage));
        }
        // stage i again from its members' current settings.  Only absorbed neighbours changed: the live filter takes their new programs and keeps its state
        // (a gain step by tag in front of a FIR: the history holds the old gain's samples, as on the host); otherwise the group is made anew
        auto rebuild = [chain, info, make_members](std::size_t stage, const std::vector<bool>& dirty, Stage* live) -> std::unique_ptr<Stage> {
            const GroupInfo& gi   = info[stage];
            std::size_t      base = 0; // index of the group's first member in the run's member list (groups are contiguous and in order)
            for (std::size_t k = 0; k < stage; ++k) base += info[k].count;
            auto made = make_members(gi.first, gi.count);
            if (made.size() != gi.count) throw std::runtime_error("device run: a member lost its device kernel");
            const bool anchored = gi.anchor != static_cast<std::size_t>(-1) && gi.count > 1;
            if (anchored && live && !dirty[base + (gi.anchor - gi.first)]) {
                live->clear_absorbed();
                bool ok = true;
                for (std::size_t m = gi.anchor - gi.first; m-- > 0 && ok;) ok = made[m]->program() && live->absorb(*made[m]->program(), true);
                for (std::size_t m = gi.anchor - gi.first + 1; m < gi.count && ok; ++m) ok = made[m]->program() && live->absorb(*made[m]->program(), false);
                if (ok) return nullptr;
            }
            return fuse_to_one(std::move(made));
        };
        if (run_edge_items) {
            (void)chain.front()->input_edges()[0]->ensure_capacity(run_edge_items);
            (void)chain.back()->output_edges()[0]->ensure_capacity(run_edge_items);
        }
        auto  run = std::make_unique<DeviceRun>(std::move(fused), chain.front()->input_edges()[0], chain.back()->output_edges()[0], chain.front()->compute_domain());
        auto* ref = run.get();
        run->set_members(std::move(members), std::move(rebuild));
        std::vector<std::unique_ptr<BlockModel>> kept;
        bool                                     placed = false;
        for (auto& bp : blocks) {
            const bool member = std::find(chain.begin(), chain.end(), bp.get()) != chain.end();
            if (member && !placed) { kept.push_back(std::move(run)); placed = true; }
            if (member) g.retired().push_back(std::move(bp)); // keep the objects alive: callers hold references to them
            else kept.push_back(std::move(bp));
        }
        blocks = std::move(kept);
        runs.push_back(ref);
    }
    return runs;
}

// ---------------------------------------------------------------------------------------------- sharded graphs: the cross-device combiner edge
// A flowgraph shards across the GPUs of a node only along independent branches -- one SDR channel per GPU -- and the placement is what the
// graph already says: every block of a branch carries compute_domain "gpu:hip:i" (ComputeDomain.hpp:47-100; EdgeParameters.domain, BlockModel.hpp:64-72).  One
// process per GPU: Shard is this process' place among them.  plan_sharded() rewrites
//     source_c -> fir_filter<complex<float>> -> PowerSpectrum -> Add<float>.in[c]   (c = 0 .. n-1, branch c on gpu:hip:d_c)   Add.out -> ...
// into ONE FanInRun per combiner: the branches with d_c mod n_ranks == rank stay (their samples enter THIS device, all of them in one launch whose store
// epilogue is the local part of math::Add: gr4hip_chain_process_multi), the other branches leave this process' schedule together with their sources, and the
// edge between the devices -- the combiner's inputs that live elsewhere -- is an RCCL collective over xGMI queued on the run's stream (gr4hip_fanin_*): every
// rank publishes the all-channel sum on the combiner's output edge (all_reduce), or its shard of the frames (reduce_scatter).
// All ranks run the same graph description on streams of the same length; an exchange covers `frames_per_exchange` frames (the last one what is left).
struct Shard {
    int             rank = 0, n_ranks = 1;
    gr4hip_fanin_t* comm = nullptr;      // null with n_ranks == 1: no collective at all; non-null: the collective runs even on one rank
    bool            scatter = false;     // true: every rank publishes only its 1 / n_ranks of each exchange's frames (reduce_scatter) instead of the whole sum
    std::size_t     frames_per_exchange = 0; // 0: as many frames as the branches' input edges hold (plan_sharded's run_edge_items / fftSize): an exchange then drains the edges,
                                             // which never have to move unread samples to make room (host-fed, 4 Mi-sample edges: 6.0 Gsamples/s drained per exchange; 64-frame exchanges 5.7 with the two-slab pipeline, 2.5 before it)
};

class FanInRun final : public BlockModel {
    struct Branch {
        std::shared_ptr<EdgeBufferBase> in;
        gr4hip_chain_t*                 chain = nullptr;
    };
    // One exchange in flight (round 5: TWO of them, the pipeline bench.py has had since round 3).  Its samples are copied on the copy stream, the ONE launch of all
    // local channels runs on the compute stream behind that copy, the collective and the copy of the sum run on the exchange stream behind the launch: the exchange of
    // launch c runs beside launch c + 1 (and beside the host's refill of the input edges), the host waits only for the copy of its OWN input spans before it hands them
    // back, and for the oldest exchange when both slabs are taken.
    struct Slab {
        std::deque<DevBuf> d_in, h_in; // per local branch
        DevBuf             d_partial{false}, d_sum{false}, h_out{true};
        gr4hip_event_t     ev_in = nullptr, ev_launch = nullptr, ev_done = nullptr;
        std::size_t        n_out = 0, exchange = 0;
        void*              direct = nullptr; // the output edge's own (page-locked) storage, reserved in order
        property_map       fwd;
        bool               busy = false;
    };
    std::deque<Branch>              _branches;
    std::shared_ptr<EdgeBufferBase> _out;
    Shard                           _shard;
    std::size_t                     _N, _n_total;
    ComputeDomain                   _domain;
    gr4hip_stream_t                 _s = nullptr, _s_copy = nullptr, _s_exchange = nullptr;
    static constexpr std::size_t    kSlabs = 2;
    std::array<Slab, kSlabs>        _slabs;
    std::size_t                     _oldest = 0, _in_flight = 0, _pending_staged = 0; // _pending_staged: output items of exchanges in flight that will be COPIED into the output edge
    std::string                     _name;
    std::size_t                     _launches = 0, _exchanges = 0, _tags_forwarded = 0, _overlapped = 0;
    bool                            _stalled = false; // an exchange timed out: the streams are abandoned (never synchronised again)

    // waits for `ev`; with a communicator the wait is bounded (other processes are involved): a diagnostic instead of a hung rank
    void wait_bounded(gr4hip_event_t ev, std::size_t exchange) {
        if (!_shard.comm || options().collective_timeout_s <= 0) { check(gr4hip_event_synchronize(ev), "event synchronize"); return; }
        const auto t0 = std::chrono::steady_clock::now();
        for (int done = 0;;) {
            check(gr4hip_event_query(ev, &done), "event query");
            if (done) return;
            const double waited = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
            if (waited > options().collective_timeout_s) {
                _stalled = true; // (the streams still hold the collective: nothing more is queued on them, the destructor does not wait for them)
                throw std::runtime_error("fan-in exchange " + std::to_string(exchange) + " of rank " + std::to_string(_shard.rank) + " / " + std::to_string(_shard.n_ranks) +
                                         " did not complete within " + std::to_string(static_cast<int>(options().collective_timeout_s)) + " s (a peer that never joined the collective?)");
            }
            std::this_thread::sleep_for(waited < 0.01 ? std::chrono::microseconds(20) : std::chrono::microseconds(500));
        }
    }
    // the oldest exchange has completed: its frames (and its tag) appear on the output edge
    std::size_t retire() {
        Slab& sl = _slabs[_oldest];
        if (!sl.fwd.empty()) { _out->publishTag(sl.fwd, 0); ++_tags_forwarded; sl.fwd.clear(); }
        if (sl.direct) _out->publish_reserved(sl.n_out);
        else {
            _pending_staged -= sl.n_out;
            if (void* dst = sl.n_out * 4 >= (std::size_t(2) << 20) ? _out->reserve_items(sl.n_out) : nullptr) { CopyPool::instance().copy(dst, sl.h_out.p, sl.n_out * 4); _out->publish_reserved(sl.n_out); } // large exchange into a pageable edge: the copy threads share it
            else _out->write_items(sl.h_out.p, sl.n_out);
        }
        sl.busy   = false;
        sl.direct = nullptr;
        _oldest   = (_oldest + 1) % kSlabs;
        --_in_flight;
        return sl.n_out;
    }

public:
    FanInRun(std::vector<std::pair<std::shared_ptr<EdgeBufferBase>, std::vector<float>>> local_branches, std::shared_ptr<EdgeBufferBase> out, std::size_t fftSize, int window,
             std::size_t n_total, Shard shard, ComputeDomain domain)
        : _out(std::move(out)), _shard(shard), _N(fftSize), _n_total(n_total), _domain(std::move(domain)) {
        check(gr4hip_set_device(_domain.index), "gr4hip_set_device");
        check(gr4hip_stream_create(&_s), "gr4hip_stream_create");
        check(gr4hip_stream_create(&_s_copy), "gr4hip_stream_create");
        check(gr4hip_stream_create(&_s_exchange), "gr4hip_stream_create");
        for (auto& [edge, taps] : local_branches) {
            _branches.emplace_back();
            _branches.back().in = std::move(edge);
            check(gr4hip_chain_create(&_branches.back().chain, taps.data(), taps.size(), fftSize, window, GR4HIP_CHAIN_AUTO), "gr4hip_chain_create");
            check(gr4hip_chain_set_guard_mode(_branches.back().chain, options().guard_mode), "gr4hip_chain_set_guard_mode");
        }
        for (auto& sl : _slabs) {
            for (std::size_t c = 0; c < _branches.size(); ++c) { sl.d_in.emplace_back(false); sl.h_in.emplace_back(true); }
            check(gr4hip_event_create(&sl.ev_in), "gr4hip_event_create");
            check(gr4hip_event_create(&sl.ev_launch), "gr4hip_event_create");
            check(gr4hip_event_create(&sl.ev_done), "gr4hip_event_create");
        }
        _name = "fan_in[" + std::to_string(_branches.size()) + " of " + std::to_string(n_total) + " channels on gpu:hip:" + std::to_string(_domain.index) + "]";
    }
    ~FanInRun() override {
        if (_stalled) return; // a collective that never completes still owns the streams and the buffers it was queued with: leaked on purpose, the process is going down
        for (gr4hip_stream_t s : {_s_copy, _s, _s_exchange})
            if (s) (void)gr4hip_stream_synchronize(s);
        for (auto& b : _branches) gr4hip_chain_destroy(b.chain);
        for (auto& sl : _slabs)
            for (gr4hip_event_t e : {sl.ev_in, sl.ev_launch, sl.ev_done})
                if (e) gr4hip_event_destroy(e);
        for (gr4hip_stream_t s : {_s_copy, _s, _s_exchange})
            if (s) gr4hip_stream_destroy(s);
    }
    [[nodiscard]] std::size_t local_channels() const { return _branches.size(); }
    [[nodiscard]] std::size_t launches() const { return _launches; }   // device launches of the branch kernels: ONE per exchange whatever the channel count
    [[nodiscard]] std::size_t exchanges() const { return _exchanges; } // collectives queued
    [[nodiscard]] std::size_t overlapped() const { return _overlapped; } // launches queued while an earlier exchange was still in flight
    [[nodiscard]] std::size_t tags_forwarded() const { return _tags_forwarded; }
    [[nodiscard]] std::size_t frames_per_exchange() const { return _shard.frames_per_exchange; }

    work::Result work(std::size_t requested) override {
        std::vector<std::size_t> lent_from; // branches whose input edge the copy engine is reading in place (spans lent in this call and not yet handed back)
        std::size_t              lent_n = 0;
        Slab*       queued   = nullptr;
        try {
            check(gr4hip_set_device(_domain.index), "gr4hip_set_device");
            std::size_t published = 0;
            // (1) exchanges that have completed appear on the output edge, in order; nobody waits here
            while (_in_flight) {
                int done = 0;
                check(gr4hip_event_query(_slabs[_oldest].ev_done, &done), "event query");
                if (!done) break;
                published += retire();
            }
            // (2) the next exchange, if a slab is free and the edges hold it
            bool        all_done = true;
            std::size_t avail    = std::numeric_limits<std::size_t>::max();
            for (auto& b : _branches) {
                avail    = std::min(avail, b.in->available_items());
                all_done = all_done && b.in->done();
            }
            if (_branches.empty()) { avail = 0; all_done = true; }
            std::size_t frames = std::min(avail / _N, _shard.frames_per_exchange);
            if (frames < _shard.frames_per_exchange && !all_done) frames = 0; // every rank exchanges the same frame counts: wait for a whole exchange
            if (frames && _shard.scatter && frames % static_cast<std::size_t>(_shard.n_ranks)) { // the stream's last exchange: reduce_scatter hands out whole frames, the same count to every rank
                const std::size_t left = frames % static_cast<std::size_t>(_shard.n_ranks);
                frames -= left;
                if (frames == 0) { // fewer frames than ranks are left: they cannot be shared out -- dropped, said once, and the stream ends DONE (not ERROR)
                    std::cerr << "[gr::hip] fan-in (reduce_scatter over " << _shard.n_ranks << " ranks): the last " << left << " frame(s) of the stream are not a whole share per rank and are dropped\n";
                    for (auto& b : _branches) b.in->consume_items(std::min(b.in->available_items(), left * _N));
                }
            }
            const std::size_t n_out = _shard.scatter ? frames / static_cast<std::size_t>(_shard.n_ranks) * _N : frames * _N;
            const bool        room  = _out->free_items() >= n_out + _pending_staged;
            if (frames && room && _in_flight < kSlabs) {
                Slab&             sl = _slabs[(_oldest + _in_flight) % kSlabs];
                const std::size_t n  = frames * _N;
                queued               = &sl;
                sl.n_out             = n_out;
                sl.direct            = nullptr;
                lent_n               = n;
                // the exchange's tag: every tag on the exchange's samples of every LOCAL branch merged ("gr:" keys; identical tags on several branches collapse, like
                // the merged input tag of an n-ary block, Block.hpp:1511-1530), published on the first output sample; gr:sample_rate follows the run's rate change
                // (1 : 1, or 1 / n_ranks of the frames in scatter mode).  Tags of branches that live on other ranks stay on those ranks.
                sl.fwd.clear();
                for (auto& b : _branches)
                    for (const auto& [key, value] : b.in->mergedTags(n)) {
                        if (!std::string_view(key).starts_with(GR_TAG_PREFIX)) continue;
                        const float* rate = _shard.scatter && tag::settingsKey(key) == tag::SAMPLE_RATE ? std::get_if<float>(&value) : nullptr;
                        if (rate) sl.fwd.insert_or_assign(key, *rate / static_cast<float>(_shard.n_ranks));
                        else sl.fwd.insert_or_assign(key, value);
                    }
                // samples of every local branch land in HBM on the copy stream (the copies of branch c + 1 behind those of branch c), beside the launch before this one
                std::vector<gr4hip_chain_t*> chains;
                std::vector<const void*>     ins;
                for (std::size_t c = 0; c < _branches.size(); ++c) {
                    auto&       b    = _branches[c];
                    const void* lent = b.in->lend_items(n);
                    if (lent && b.in->memory() == pinned_resource()) { // page-locked edge ("hip" provider): the copy engine reads the edge in place; the span goes back below
                        lent_from.push_back(c);
                        check(gr4hip_memcpy_h2d(sl.d_in[c].ensure(n * 8), lent, n * 8, _s_copy), "h2d");
                    } else {
                        if (lent) { // pageable edge: staged through the slab's page-locked memory by the copy threads
                            CopyPool::instance().copy(sl.h_in[c].ensure(n * 8), lent, n * 8);
                            b.in->consume_items(n);
                        } else {
                            b.in->read_items(sl.h_in[c].ensure(n * 8), n);
                        }
                        check(gr4hip_memcpy_h2d(sl.d_in[c].ensure(n * 8), sl.h_in[c].p, n * 8, _s_copy), "h2d");
                    }
                    chains.push_back(b.chain);
                    ins.push_back(sl.d_in[c].p);
                }
                check(gr4hip_event_record(sl.ev_in, _s_copy), "event record");
                check(gr4hip_stream_wait_event(_s, sl.ev_in), "stream wait");
                // ONE launch for all local channels, the local part of math::Add as its store epilogue
                std::size_t got = 0;
                check(gr4hip_chain_process_multi(chains.data(), chains.size(), ins.data(), n, nullptr, static_cast<float*>(sl.d_partial.ensure(n * 4)), &got, _s), "gr4hip_chain_process_multi");
                ++_launches;
                if (_in_flight) ++_overlapped;
                check(gr4hip_event_record(sl.ev_launch, _s), "event record");
                check(gr4hip_stream_wait_event(_s_exchange, sl.ev_launch), "stream wait");
                const float* result = static_cast<const float*>(sl.d_partial.p);
                if (_shard.comm) { // the combiner's inputs that live on other devices: one collective per exchange, beside the NEXT launch
                    float* sum = static_cast<float*>(sl.d_sum.ensure(n_out * 4));
                    if (_shard.scatter) check(gr4hip_fanin_reduce_scatter_sum_f32(_shard.comm, result, sum, n_out, _s_exchange), "gr4hip_fanin_reduce_scatter_sum_f32");
                    else check(gr4hip_fanin_all_reduce_sum_f32(_shard.comm, result, sum, n_out, _s_exchange), "gr4hip_fanin_all_reduce_sum_f32");
                    result = sum;
                    ++_exchanges;
                } else if (_shard.n_ranks != 1) {
                    throw std::runtime_error("sharded graph without a communicator");
                }
                sl.direct = _out->memory() == pinned_resource() ? _out->reserve_items(n_out) : nullptr; // a page-locked output edge takes the result copy in its own storage
                check(gr4hip_memcpy_d2h(sl.direct ? sl.direct : sl.h_out.ensure(n_out * 4), result, n_out * 4, _s_exchange), "d2h");
                check(gr4hip_event_record(sl.ev_done, _s_exchange), "event record");
                sl.exchange = _exchanges;
                sl.busy     = true;
                if (!sl.direct) _pending_staged += n_out;
                ++_in_flight;
                queued = nullptr; // (from here on the slab is the pipeline's: a failure below is the run's failure, handled as a whole)
                // the input spans go back as soon as the copy engine has read them (the launch and the exchange are still running): the sources refill beside the device
                if (!lent_from.empty()) check(gr4hip_event_synchronize(sl.ev_in), "event synchronize");
                for (const std::size_t c : lent_from) _branches[c].in->consume_items(n);
                lent_from.clear();
                return {requested, published + n, work::Status::OK};
            }
            if (published) return {requested, published, work::Status::OK};
            // (3) nothing could be queued and nothing had completed: the oldest exchange is what everybody is waiting for
            if (_in_flight) {
                wait_bounded(_slabs[_oldest].ev_done, _slabs[_oldest].exchange);
                return {requested, retire(), work::Status::OK};
            }
            if (frames && !room) return {requested, 0, work::Status::INSUFFICIENT_OUTPUT_ITEMS};
            if (all_done) { _out->producer_done = true; return {requested, 0, work::Status::DONE}; }
            return {requested, 0, work::Status::INSUFFICIENT_INPUT_ITEMS};
        } catch (const std::exception& e) {
            std::cerr << "[gr::hip] fan-in run failed: " << e.what() << "\n";
            if (_stalled) return {requested, 0, work::Status::ERROR}; // (nothing can be taken back from streams that do not drain)
            for (gr4hip_stream_t s : {_s_copy, _s, _s_exchange}) (void)gr4hip_stream_synchronize(s); // nothing of the failed exchange may still read a lent span
            for (const std::size_t c : lent_from) _branches[c].in->unlend_items(lent_n);
            std::size_t held = queued && queued->direct ? queued->n_out : 0;
            for (auto& sl : _slabs)
                if (sl.busy && sl.direct) held += sl.n_out;
            if (held) _out->unreserve_items(held);
            //  the pipeline starts over: nothing that was in flight is published later -- its reservations are gone, a retire() after this error would
            // publish storage that is no longer reserved
            for (auto& sl : _slabs) { sl.busy = false; sl.direct = nullptr; sl.fwd.clear(); }
            _in_flight = 0; _pending_staged = 0; _oldest = 0;
            return {requested, 0, work::Status::ERROR};
        }
    }
    std::string_view     name() const override { return _name; }
    std::string_view     type_name() const override { return "gr::hip::FanInRun"; }
    const ComputeDomain& compute_domain() const override { return _domain; }
    void*                raw() override { return this; }
    std::type_index      port_type(std::string_view) override { return typeid(void); }
    std::shared_ptr<EdgeBufferBase> make_edge(std::string_view, std::size_t, std::pmr::memory_resource*) override { return nullptr; }
    bool attach_input(std::string_view, std::shared_ptr<EdgeBufferBase>) override { return false; }
    std::vector<std::shared_ptr<EdgeBufferBase>> input_edges() override {
        std::vector<std::shared_ptr<EdgeBufferBase>> v;
        for (auto& b : _branches) v.push_back(b.in);
        return v;
    }
    std::vector<std::shared_ptr<EdgeBufferBase>> output_edges() override { return {_out}; }
};

// Finds every combiner  Add<float>(n inputs)  all of whose inputs are  fir_filter<complex<float>> -> PowerSpectrum  branches on "gpu:hip:i" domains and replaces
// combiner + branches by one FanInRun for this rank (see above).  Returns the runs it created; graphs without such a combiner are left alone.
inline std::vector<FanInRun*> plan_sharded(Graph& g, const Shard& shard, std::size_t run_edge_items = std::size_t(1) << 22) {
    auto& blocks = g.blocks();
    const auto producer = [&](const std::shared_ptr<EdgeBufferBase>& e) -> BlockModel* {
        for (auto& b : blocks)
            for (auto& out : b->output_edges())
                if (out == e) return b.get();
        return nullptr;
    };
    const auto hip_device = [](BlockModel& b) { const auto& d = b.compute_domain(); return d.is_device() && (d.backend.empty() || d.backend == "hip"); };
    std::vector<FanInRun*> runs;
    for (std::size_t pos = 0; pos < blocks.size(); ++pos) {
        BlockModel* add = blocks[pos].get();
        const std::string_view tn = add->type_name();
        if (tn.find("MathOpMultiPortImpl") == std::string_view::npos && tn.find("math::Add") == std::string_view::npos) continue;
        if (add->port_type("out") != typeid(float)) continue;
        const auto ins = add->input_edges();
        if (ins.size() < 1 || add->output_edges().size() != 1) continue;
        struct Found { BlockModel *fir, *spec; std::vector<BlockModel*> upstream; std::vector<float> taps; std::size_t N; int window; int device; };
        std::vector<Found> found;
        bool               ok = true;
        for (auto& e : ins) {
            BlockModel* spec = e ? producer(e) : nullptr;
            BlockModel* fir  = spec && spec->input_edges().size() == 1 ? producer(spec->input_edges()[0]) : nullptr;
            if (!spec || !fir || !hip_device(*spec) || !hip_device(*fir) || fir->input_edges().size() != 1) { ok = false; break; }
            auto ss = std::dynamic_pointer_cast<PowerSpectrumStage>(std::static_pointer_cast<Stage>(spec->make_device_stage()));
            auto fs = std::dynamic_pointer_cast<FirStage<std::complex<float>>>(std::static_pointer_cast<Stage>(fir->make_device_stage()));
            if (!ss || !fs) { ok = false; break; }
            Found f{fir, spec, {}, fs->taps, ss->N, ss->window, fir->compute_domain().index};
            // everything upstream of the branch feeds only this branch: it leaves the schedule with it when the branch lives on another rank
            std::vector<BlockModel*> todo{fir};
            while (!todo.empty()) {
                BlockModel* b = todo.back();
                todo.pop_back();
                for (auto& ie : b->input_edges())
                    if (BlockModel* up = ie ? producer(ie) : nullptr) { f.upstream.push_back(up); todo.push_back(up); }
            }
            found.push_back(std::move(f));
        }
        if (!ok || found.empty()) continue;
        for (auto& f : found) ok = ok && f.N == found[0].N && f.window == found[0].window;
        if (!ok) continue;
        std::vector<std::pair<std::shared_ptr<EdgeBufferBase>, std::vector<float>>> local;
        std::vector<BlockModel*> retire;
        int                      device = -1;
        for (auto& f : found) {
            const bool mine = f.device % shard.n_ranks == shard.rank;
            if (mine) {
                if (run_edge_items) (void)f.fir->input_edges()[0]->ensure_capacity(run_edge_items);
                local.emplace_back(f.fir->input_edges()[0], f.taps);
                if (device < 0) device = f.device;
            } else {
                for (auto* up : f.upstream) retire.push_back(up); // the other ranks' sources are not this process' business
            }
            retire.push_back(f.fir);
            retire.push_back(f.spec);
        }
        retire.push_back(add);
        if (local.empty()) throw std::runtime_error("plan_sharded: rank " + std::to_string(shard.rank) + " of " + std::to_string(shard.n_ranks) + " owns no branch of the combiner (more ranks than gpu:hip devices in the graph): it could not take part in the exchanges");
        ComputeDomain dom = found[0].fir->compute_domain();
        if (device >= 0) dom.index = device;
        Shard sh = shard;
        if (sh.frames_per_exchange == 0) { // as much as the edges around the run hold (the same graph, hence the same number, on every rank)
            std::size_t cap = std::numeric_limits<std::size_t>::max();
            for (auto& [edge, taps] : local) cap = std::min(cap, edge->capacity_items());
            sh.frames_per_exchange = std::max<std::size_t>(1, cap / found[0].N);
        }
        {   // the combiner's output edge takes a whole exchange
            auto              out  = add->output_edges()[0];
            const std::size_t need = sh.frames_per_exchange * found[0].N / (sh.scatter ? static_cast<std::size_t>(sh.n_ranks) : 1);
            if (!out->ensure_capacity(need) && out->capacity_items() < need)
                sh.frames_per_exchange = std::max<std::size_t>(1, out->capacity_items() * (sh.scatter ? static_cast<std::size_t>(sh.n_ranks) : 1) / found[0].N);
        }
        if (sh.scatter) sh.frames_per_exchange = std::max<std::size_t>(1, sh.frames_per_exchange / static_cast<std::size_t>(sh.n_ranks)) * static_cast<std::size_t>(sh.n_ranks);
        auto  run = std::make_unique<FanInRun>(std::move(local), add->output_edges()[0], found[0].N, found[0].window, found.size(), sh, dom);
        auto* ref = run.get();
        std::vector<std::unique_ptr<BlockModel>> kept;
        for (auto& bp : blocks) {
            if (bp.get() == add) kept.push_back(std::move(run));
            if (std::find(retire.begin(), retire.end(), bp.get()) != retire.end()) g.retired().push_back(std::move(bp)); // (callers hold references to the blocks)
            else kept.push_back(std::move(bp));
        }
        blocks = std::move(kept);
        runs.push_back(ref);
        pos = static_cast<std::size_t>(-1); // the block list changed: start over (combiners already replaced are FanInRuns now)
    }
    return runs;
}

} // namespace gr::hip
