/*
 * gr4hip.h -- C-ABI of the MI355X (gfx950) kernel library behind GNU Radio 4's block API.
 *
 * This is the drop-in boundary: plain pointers, sizes and opaque handles; no C++ or torch
 * types; nothing throws across it.  Each entry point replaces the per-block processOne/processBulk arithmetic
 * that gr::Block<>::dispatchProcessing (core/include/gnuradio-4.0/Block.hpp:1848-1917, device seam :1855-1862)
 * would run on the CPU.  The reference has no FFI of its own for this path -- the seam is the inert
 * `compute_domain == "gpu:hip"` hook -- so this header IS the binding a maintainer adds there
 * (INTEGRATION.md shows the stub).  All citations are relative to /root/reference.
 *
 * Conventions
 *   - every function returns gr4hip_status; values -100..0 are gr::work::Status
 *     (core/include/gnuradio-4.0/WorkStatus.hpp:12-41), library errors are < -100.
 *   - `d_*` pointers are device (HBM) pointers on the current device; `stream` is a hipStream_t passed as void*
 *     (NULL = default stream).  Calls are asynchronous on that stream: spans are borrowed until the stream
 *     reaches the end of the enqueued work (Block.hpp:1838-1846 publish/consume ordering is the caller's).
 *   - complex samples are interleaved {re, im}; dtype ids are gr4hip_dtype.
 *   - state that the reference keeps in block members (HistoryBuffer, _accumulated_phase, twiddles, windows)
 *     lives in opaque handles created/destroyed by *_create / *_destroy.
 *   - LIFECYCLE CALLS ARE STREAM-ORDERED WITH THE DATA.  *_reset, *_set_taps, gr4hip_fir_set_prologue / _epilogue, *_set_algo and *_set_guard_mode return at once
 *     and never touch device memory: they note what the handle's device state has to become, and the NEXT *_process call on the handle applies it on the stream
 *     IT is given (hipMemsetAsync / hipMemcpyAsync / a small kernel), in front of its own launches -- hence behind every launch that stream still has in flight
 *     for the handle.  This is the reference's contract: reset() and settingsChanged() run on the block's own worker between two work() calls
 *     (Block.hpp:606, 916-917, 1296; one worker per job list, Scheduler.hpp:1938-1951), so they are ordered with the block's samples by construction.  It holds
 *     on hipStreamNonBlocking streams (gr4hip_stream_create makes those), which the NULL stream does not order against, and needs no device-wide wait.  A caller
 *     that moves a handle from one stream to another orders the two streams itself (gr4hip_event_record + gr4hip_stream_wait_event), as it must for the carried
 *     history anyway.  After a settings change the first process call may hold the HOST until that stream has drained: the new tables go up from pageable memory,
 *     stream-ordered.  Calls that hand a device value back to the host (gr4hip_rotator_phase under the recurrence, gr4hip_iir_status; gr4hip_iir_set_algo and
 *     gr4hip_rotator_set_algo when they have to measure or read something) wait for what they need and say so.  *_create uploads block and are complete on return.
 *   - one handle is driven by one thread at a time (Scheduler.hpp:1938-1951: job lists are disjoint).
 *   - PARITY CONTRACT (stated here once; tests/test_gpu_parity.py::_rel is this formula and every float test uses it): integer, byte and copy results are
 *     bit-exact.  A float32 result y against the float64 evaluation t of the same blocks on the same input satisfies
 *         max_k |y_k - t_k| / max(|t_k|, rms(t)) <= 1e-5
 *     -- the relative error of every value above the rms level of the OUTPUT, the rms-normalised absolute error of every value below it (point-wise relative
 *     error is meaningless at spectral zeros; an all-rms normalisation would ask for better than float32 epsilon on the dominant bin of a quadratic output).
 *     Where the reference's own float32 arithmetic cannot meet that bound (a rejected signal far above the output, an ill-conditioned IIR cascade), the bound is
 *     the reference's float32 error on the same input -- the error of its sums evaluated in float32 in ITS order (time_domain_filter.hpp:44-47, iir section by
 *     section) against float64 -- with a factor of ONE, at every shape (tests and fuzzers compare with the oracle's restatement of exactly that arithmetic, never
 *     with another device kernel).  What keeps the device there: every FIR / decimator kernel answers to one guard -- a segment whose output power is more than
 *     21 dB below what white noise of its input power would pass is evaluated again with float64 products and sums (fir_exact.hip; error ~6e-8 of the output
 *     whatever the signal) --, the fused chain marks such frames and evaluates them again in the time domain behind its launch, and an ill-conditioned IIR
 *     cascade runs on GR4HIP_IIR_SEQUENTIAL_F32.
 */
#ifndef GR4HIP_H
#define GR4HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GR4HIP_ABI_VERSION 1 /* cf. gr_plugin_base::abiVersion() == 1, core/include/gnuradio-4.0/Plugin.hpp:20-33 */

typedef enum {
    GR4HIP_OK                  = 0,    /* work::Status::OK */
    GR4HIP_DONE                = -1,   /* work::Status::DONE */
    GR4HIP_INSUFFICIENT_INPUT  = -2,   /* work::Status::INSUFFICIENT_INPUT_ITEMS */
    GR4HIP_INSUFFICIENT_OUTPUT = -3,   /* work::Status::INSUFFICIENT_OUTPUT_ITEMS */
    GR4HIP_ERROR               = -100, /* work::Status::ERROR */
    GR4HIP_INVALID_ARGUMENT    = -101,
    GR4HIP_RUNTIME_ERROR       = -102, /* a HIP runtime call failed; see gr4hip_last_error() */
    GR4HIP_UNSUPPORTED         = -103, /* valid request the device path does not implement (caller keeps the CPU path) */
    GR4HIP_NO_DEVICE           = -104
} gr4hip_status;

typedef enum { /* element types of the reference's block registrations (Math.hpp:25-28, time_domain_filter.hpp:20,213) */
    GR4HIP_U8 = 0, GR4HIP_U16, GR4HIP_U32, GR4HIP_U64, GR4HIP_I8, GR4HIP_I16, GR4HIP_I32, GR4HIP_I64,
    GR4HIP_F32, GR4HIP_F64, GR4HIP_C32, GR4HIP_C64,
    /* gr::UncertainValue<float | double> (meta/.../UncertainValue.hpp:34): an element is the pair {value, uncertainty}, 8 / 16 bytes, laid out like the struct.
     * Taken by the math blocks (gr4hip_math_const / gr4hip_math_nary: both operands carry an uncertainty, uncorrelated propagation as the reference's operators
     * write it, UncertainValue.hpp:121-250: a +- b -> hypot(ua, ub); a * b -> hypot(a ub, b ua); a / b -> hypot(ua / b, ub a / b^2)), by gr4hip_decimate and the
     * ring buffers (any element size).  Not a filter / transform sample type here. */
    GR4HIP_UF32, GR4HIP_UF64
} gr4hip_dtype;

typedef enum { GR4HIP_ADD = 0, GR4HIP_SUB, GR4HIP_MUL, GR4HIP_DIV } gr4hip_op; /* std::plus/minus/multiplies/divides */

typedef enum { GR4HIP_GUARD_STRICT = 0, GR4HIP_GUARD_DEFERRED = 1, GR4HIP_GUARD_OFF = 2 } gr4hip_guard_mode; /* dynamic-range guard: gr4hip_chain_set_guard_mode */

typedef enum { /* gr::filter::IIRForm (time_domain_filter.hpp:50-55) == gr::filter::Form (FilterTool.hpp:107-112) */
    GR4HIP_DF_I = 0, GR4HIP_DF_II, GR4HIP_DF_I_TRANSPOSED, GR4HIP_DF_II_TRANSPOSED
} gr4hip_iir_form;

typedef enum { /* gr::algorithm::window::Type (algorithm/.../fourier/window.hpp:35) */
    GR4HIP_WIN_NONE = 0, GR4HIP_WIN_RECTANGULAR, GR4HIP_WIN_HAMMING, GR4HIP_WIN_HANN, GR4HIP_WIN_HANNEXP,
    GR4HIP_WIN_BLACKMAN, GR4HIP_WIN_NUTTALL, GR4HIP_WIN_BLACKMANHARRIS, GR4HIP_WIN_BLACKMANNUTTALL,
    GR4HIP_WIN_FLATTOP, GR4HIP_WIN_EXPONENTIAL, GR4HIP_WIN_KAISER
} gr4hip_window;

enum { /* FFT block output options (blocks/fourier/.../fft.hpp:103-105) */
    GR4HIP_FFT_OUTPUT_IN_DB  = 1,
    GR4HIP_FFT_OUTPUT_IN_DEG = 2,
    GR4HIP_FFT_UNWRAP_PHASE  = 4
};

typedef enum { /* how the FIR->FFT->mag2 chain is executed */
    GR4HIP_CHAIN_AUTO = 0,
    GR4HIP_CHAIN_UNFUSED,  /* fir kernel -> HBM -> fft+mag2 kernel (any window, any size the FFT block supports: what AUTO takes beyond 8192 points, at sizes that are no power of two
                              and past 256 taps).  Since round 6 the filter runs on float32 products here too (= GR4HIP_CHAIN_TIME_DOMAIN): the 22-bit products of the f16 direct form err
                              COHERENTLY on a tone, and the transform behind the filter gathers that into one bin -- up to 4e-5 of |Y|^2 where a tone 15 dB above the noise is removed by
                              60 dB (tools/dbg/pair_coherent.py), which no power statistic of the filter can see.  44 Gsamples/s at 256 taps x 16384 points (130 on the f16 form) */
    GR4HIP_CHAIN_FUSED_TD, /* one launch: direct-form FIR on the matrix pipe -> window -> FFT -> mag2 (fft_size 256 .. 4096, <= 256 taps); AUTO takes it for <= 64 taps */
    GR4HIP_CHAIN_FUSED_FD, /* one launch, frequency-domain FIR (circular convolution + exact tail correction) + FFT + mag2;
                              fft_size 256 ... 8192 (power of two), <= 256 taps, any window */
    GR4HIP_CHAIN_TIME_DOMAIN /* direct-form FIR kernel with float32 products (GR4HIP_FIR_TIME_DOMAIN_F32) -> HBM -> fft+mag2 kernel: the reference's arithmetic; for
                              inputs whose out-of-band content dwarfs the filtered output (see gr4hip_fir_set_algo), and where the dynamic-range guard sends a stream */
} gr4hip_chain_algo_t;

typedef struct gr4hip_ewise gr4hip_ewise_t; /* a run of per-sample blocks (gr4hip_ewise_* below) */
typedef void* gr4hip_stream_t; /* hipStream_t */
typedef void* gr4hip_event_t;  /* hipEvent_t  */

/* ------------------------------------------------------------------------------------------------ runtime */
int         gr4hip_abi_version(void);
const char* gr4hip_last_error(void); /* thread-local text of the last failure */
/* Developer switches: which of two kernels that compute the same thing serves a call (the tests compare them; tools/ time them).  Each is read from the
 * environment variable of the same name ONCE, when the library is first used, and can be changed afterwards with this call (atomic; process-wide).
 * Nothing here changes the meaning of a call -- choices that do (exact float32 FIR arithmetic, the rotator's phase recurrence, the chain's algorithm and
 * guard) are per-handle settings: gr4hip_fir_set_algo, gr4hip_rotator_set_algo, gr4hip_chain_create / gr4hip_chain_set_guard_mode.
 * Names: GR4HIP_FIR_NO_BF16X3, GR4HIP_FIR_NO_DECIM_FD, GR4HIP_IIR_THREE_PASS, GR4HIP_IIR_LOOKBACK, GR4HIP_IIR_NO_SPLIT, GR4HIP_FFT_BLUESTEIN_PIPELINE,
 * GR4HIP_FFT_NO_PIPELINE, GR4HIP_ROTATOR_LEAP, GR4HIP_ROTATOR_WALK, GR4HIP_CHAIN16, GR4HIP_FFT_SMOOTH_RUNTIME, GR4HIP_EWISE_NO_DIV_RCP. */
int gr4hip_developer_switch(const char* name, int value);
const char* gr4hip_status_string(int status);
int         gr4hip_device_count(int* count);
int         gr4hip_set_device(int index); /* ComputeDomain "gpu:hip:<index>" (ComputeDomain.hpp:47-100) */
int         gr4hip_get_device(int* index);
int         gr4hip_device_name(int index, char* buf, size_t buflen);

/* "hip" memory provider (the ComputeRegistry::register_provider("hip", fn) resources, ComputeDomain.hpp:105-173) */
int gr4hip_malloc(void** d_ptr, size_t bytes);
int gr4hip_free(void* d_ptr);
int gr4hip_malloc_host(void** h_ptr, size_t bytes); /* pinned host staging */
int gr4hip_free_host(void* h_ptr);
/* A page-locked host RING (round 5): `bytes` of storage (a multiple of the page size) mapped TWICE back to back -- the double mapping of the reference's CircularBuffer
 * (core/include/gnuradio-4.0/CircularBuffer.hpp:75-172: memfd + two mmaps) on the host side of the link, registered with the runtime: base[0 .. 2 bytes) is addressable,
 * base[i] and base[i + bytes] are the same byte, so a span that wraps the physical end is contiguous in virtual memory and the copy engines read / write it in place at the
 * link's rate (measured: 56.2 GB/s inside the mapping, across its end, and from hipHostMalloc memory alike).  What a CPU-domain edge that feeds the device is made of:
 * nothing is ever moved to the front of such an edge. */
int gr4hip_host_ring_create(void** base, size_t bytes);
int gr4hip_host_ring_destroy(void* base, size_t bytes);
int gr4hip_memcpy_h2d(void* d_dst, const void* h_src, size_t bytes, gr4hip_stream_t stream);
int gr4hip_memcpy_d2h(void* h_dst, const void* d_src, size_t bytes, gr4hip_stream_t stream);
int gr4hip_memcpy_d2d(void* d_dst, const void* d_src, size_t bytes, gr4hip_stream_t stream);
int gr4hip_memset(void* d_dst, int value, size_t bytes, gr4hip_stream_t stream);
int gr4hip_stream_create(gr4hip_stream_t* stream);
int gr4hip_stream_destroy(gr4hip_stream_t stream);
int gr4hip_stream_synchronize(gr4hip_stream_t stream);
int gr4hip_stream_query(gr4hip_stream_t stream, int* idle); /* *idle = 1: everything queued on the stream has finished (hipStreamQuery; never waits) -- one call instead of an event record + query per copy for a caller that keeps one copy in flight per stream (gr::hip::DeviceRun's pieces) */
int gr4hip_event_create(gr4hip_event_t* ev);
int gr4hip_event_destroy(gr4hip_event_t ev);
int gr4hip_event_record(gr4hip_event_t ev, gr4hip_stream_t stream);
int gr4hip_stream_wait_event(gr4hip_stream_t stream, gr4hip_event_t ev); /* work queued on `stream` after this call starts once `ev` has fired */
int gr4hip_event_synchronize(gr4hip_event_t ev);
int gr4hip_event_query(gr4hip_event_t ev, int* done);
int gr4hip_event_elapsed_ms(gr4hip_event_t start, gr4hip_event_t stop, float* ms);

/* GPU-resident double-mapped ring: the device analogue of CircularBuffer's memfd_create + 2x mmap
 * (core/include/gnuradio-4.0/CircularBuffer.hpp:75-172, 191-236): [base, base+size) and [base+size, base+2*size)
 * alias the same HBM, so a reader/writer span never has to be split at the wrap point. */
typedef struct gr4hip_ring gr4hip_ring_t;
int gr4hip_ring_create(gr4hip_ring_t** ring, size_t min_bytes); /* size rounded up to the VMM granularity */
int gr4hip_ring_destroy(gr4hip_ring_t* ring);
int gr4hip_ring_base(const gr4hip_ring_t* ring, void** d_base);
int gr4hip_ring_size(const gr4hip_ring_t* ring, size_t* bytes);

/* ------------------------------------------------------------------------------------------------ a1/a2/a5/a6
 * gr::filter::fir_filter<T>::processOne (blocks/filter/.../time_domain_filter.hpp:44-47):
 *   y[n] = sum_k b[k] x[n-k], zero initial history, history carried across calls (HistoryBuffer.hpp:130-139).
 * dtype F32 (registered) or C32 (complex data x real taps).  decim > 1 gives the
 * BasicFilterProto decimating processBulk (:190-204): output m == y[m*decim]; n_in must be a multiple of decim
 * (the reference guarantees it through input_chunk_size = decimate, :166-168). */
/* Decimate by 8 with 97 .. 1025 taps (float, long 16-byte-aligned spans: BASELINE configs[2]'s filter), by 16 with 33 .. 897 and by 32 with 33 .. 641 taps (complex data: by 8 with 64 .. 513, by 16 with 33 .. 449, by 32 with 33 .. 321 taps) run since late round 4 on the f16 matrix pipe with the arithmetic and
 * the safeguards described below for the non-decimating filters (csrc/fir_decim_f16.hip: block exponent per segment of 1024 outputs, per-segment verdict, second evaluation with
 * three-term f16 products, float32 sums for segments with a non-finite sample or a spread beyond 2^28): error relative to the OUTPUT, the call asynchronous.  The
 * frequency-domain kernel of rounds 1-3 (error floor relative to the input, strict host-side guard) is what GR4HIP_FIR_IIR_ONE_LAUNCH fuses the cascade into. */
/* Non-finite and near-FLT_MAX samples (pinned by tests/test_gpu_parity.py::test_fir_non_finite_samples / test_fir_f16_kernel_outliers_and_non_finite_samples /
 * test_chain_non_finite_samples).  The reference's transform_reduce gives +-Inf / NaN on exactly the ntaps outputs whose window contains such a sample.  With the
 * default algorithm:
 *  - 33 .. 256 taps (float and complex; as slices of 256 taps that add into y: float up to 3840, complex up to 1792 taps) on long 16-byte-aligned spans evaluate the products on the f16 matrix pipe, samples and taps split
 *    into two f16 terms under a block exponent per segment of 4096 (complex: 2048) outputs (csrc/fir_f16.hip; same 1e-5 parity bar).  A segment that holds a
 *    non-finite sample is evaluated as plain float32 sums instead: the reference's classes (+Inf, -Inf, NaN) on exactly its outputs; a finite outlier more than
 *    2^28 above the segment's ordinary level (1e30 or 3.4e38 beside unit-power samples, a burst that ends inside the segment) sends its segment to float32 products on the f32 matrix pipe, so the
 *    samples beside it keep their accuracy.  (GR4HIP_FIR_TIME_DOMAIN_BF16X3, the three-term bf16 kernels of rounds 2-3: every output the reference makes
 *    non-finite is non-finite, as NaN, the reach is the kernel's 32-sample-granular window -- at most 15 outputs earlier and 46 later than the reference's --,
 *    and a finite sample above bf16's largest value, 3.39e38, counts as infinite.)
 *  - complex data, 97 .. 256 taps, spans of >= 64 x 8192 samples take a fast-convolution kernel: a non-finite sample reaches every output of its 8192-sample
 *    block (and of the next block when it lies within the block's last 255 samples) instead of the next ntaps outputs.
 * gr4hip_fir_set_algo(GR4HIP_FIR_EXACT_F32) selects, per handle, kernels whose classes (+Inf, -Inf, NaN) and reach are the reference's exactly. */
typedef struct gr4hip_fir gr4hip_fir_t;
int gr4hip_fir_create(gr4hip_fir_t** fir, int dtype, const float* h_taps, size_t ntaps, size_t decim);
int gr4hip_fir_set_taps(gr4hip_fir_t* fir, const float* h_taps, size_t ntaps); /* settingsChanged (:38-42): history is kept */
int gr4hip_fir_reset(gr4hip_fir_t* fir);
/* GR4HIP_FIR_AUTO (default): since round 4 complex data with 33 .. 256 taps takes the f16 direct form on 16-byte-aligned spans (error relative to the output, see
 * below); long complex spans that are only 8-byte aligned (97 .. 256 taps) take the fast-convolution kernel.  Its float32 error floor is ~2e-6 of the INPUT rms per output sample
 * (three transforms' worth of rounding), the direct form's ~2e-7: when out-of-band signals that the filter removes are much stronger than what it
 * passes, GR4HIP_FIR_TIME_DOMAIN keeps the error relative to the OUTPUT inside the 1e-5 parity bar (the reference's own arithmetic, 1024 flop/sample). */
/* FIR_AUTO carries the same dynamic-range guard as GR4HIP_CHAIN_AUTO (gr4hip_chain_last_power_ratio below): the first fast convolution of a stream is probed
 * on eight frames, later ones are watched through the powers every launch measures (every frame judged by itself), and below an output / input power ratio of
 * 0.04 the direct form takes over. */
/* Accuracy of the direct form on the matrix pipes (the default for 33 .. 3840 taps -- complex: .. 1792 -- and the decimators; GR4HIP_FIR_TIME_DOMAIN for complex data).  Two-term f16
 * splits under a per-segment block exponent, three products per tap (everything above 2^-22 of a product; a correctly rounded float32 product carries 2^-25): on
 * ordinary input the error against float64 is that of a float32 sum (3e-7 .. 6e-7).  The error is relative to the PRODUCTS, so it shows against the OUTPUT when the
 * filter removes nearly all it is given -- like the reference's own float32 sum, whatever its order.  ONE guard for every kernel gr4hip_fir_process can take (round 5):
 * each segment's output power P_y is compared with its input power P_x, and a segment with D P_y < (sum b^2 / 128) P_x (21 dB more rejected than white noise would lose;
 * the split products are then at <= 6e-6 of the output; the f16 decimators mark at sum b^2 / 64 since round 6: a tone in a long filter's transition band left 9e-6) is marked and evaluated again with float64 products and sums, rounded once -- by fir_exact_kernel on the FP64
 * matrix pipe behind the launch (same stream, no host), inside the workgroup for the register-window kernel, with the filter's load / store programs applied where it
 * carries any.  Result: within 1e-5 of float64 on every stream tools/tone_ratio.py, tools/fuzz_fir_f16.py and the tests could construct (rejected tones up to 70 dB
 * above what passes: 6e-8), where the reference's float32 sum itself is at 1e-5 .. 1e-3.  A stream in which EVERY segment is marked runs at the FP64 matrix pipe's
 * rate (measured, profiles/r05_rejected_stream_rates.txt: float 256 taps 264 Gsamples/s against 518 unmarked, complex 143 against 178, decimate-by-8 with 1024 taps 417 against 749).  gr4hip_fir_set_guard_mode(GR4HIP_GUARD_OFF) switches the verdict off (the kernels' own products).
 * Non-finite samples: a segment whose window holds one keeps the main kernel's float32 sums (the reference's classes and reach). */
/* GR4HIP_FIR_TIME_DOMAIN_BF16X3: the three-term bf16 products of rounds 2-3 (six products per tap, everything above 2^-23 of a product, float32's exponent range
 * without a block exponent; judged like every kernel) where the default takes the f16 kernels; for complex data also "direct form" like GR4HIP_FIR_TIME_DOMAIN. */
/* GR4HIP_FIR_EXACT_F32: every product and sum in IEEE float32 (no frequency-domain kernels, no bf16 splits): the kernels whose arithmetic is the reference's
 * transform_reduce (time_domain_filter.hpp:44-47) term for term up to the order of the additions -- an infinite or NaN input sample reaches exactly the
 * ntaps outputs whose window contains it, as +-Inf / NaN (tests/test_gpu_parity.py::test_fir_non_finite_samples pins this and what the default algorithm
 * does instead).  Slower for 33 .. 1024 taps (the f32 MFMA / register-window kernels: DESIGN.md 3.2, 3.6). */
/* GR4HIP_FIR_TIME_DOMAIN_F32: the direct form with float32 products on the f32 matrix pipe (block-wise float32 sums: at least as close to float64 as the
 * reference's sequential float32 sum -- 7e-6 of the output where a rejected signal 50 dB above it leaves the float32 CPU form at 3e-5 and the three-term bf16
 * products at 1e-4), 33 .. 256 taps at about a third of the bf16 kernels' rate; Inf / NaN reach as described above.  Judged like every kernel (above). */
typedef enum { GR4HIP_FIR_AUTO = 0, GR4HIP_FIR_TIME_DOMAIN = 1, GR4HIP_FIR_EXACT_F32 = 2, GR4HIP_FIR_TIME_DOMAIN_F32 = 3, GR4HIP_FIR_TIME_DOMAIN_BF16X3 = 4 } gr4hip_fir_algo;
int gr4hip_fir_set_algo(gr4hip_fir_t* fir, int algo);
int gr4hip_fir_set_guard_mode(gr4hip_fir_t* fir, int mode); /* gr4hip_guard_mode (below), for FIR_AUTO's fast convolution of long complex spans and (GUARD_OFF) the f16 direct form's per-segment verdict; default GR4HIP_GUARD_STRICT */
int gr4hip_fir_process(gr4hip_fir_t* fir, const void* d_in, size_t n_in, void* d_out, size_t* n_out, gr4hip_stream_t stream);
int gr4hip_fir_destroy(gr4hip_fir_t* fir);

/* Interpolating FIR (BASELINE.json north_star "decimating / interpolating FIR").  The reference has no such block, only the rate declaration
 * it would carry, Resampling<1, L> (core/include/gnuradio-4.0/annotated.hpp:121-128; chunk bookkeeping Block.hpp:1576-1636), so the definition is
 * SURVEY.md Appendix A's: zero-stuff by `interp`, then fir_filter's sum at the output rate, gain `interp`:
 *   u[n] = x[n / L] if n % L == 0 else 0,   y[n] = L sum_k b[k] u[n - k]   ==   y[m L + p] = sum_q (L b[q L + p]) x[m - q]   (polyphase, evaluated)
 * n_out = n_in * interp; zero initial history, ceil(ntaps / interp) - 1 input samples carried across calls; dtype F32 or C32 (real taps).
 * Parity is pinned by the project's own float64 oracle (gr4o_fir_interp_*: literal zero-stuffing + the a1 sum), not by the reference. */
typedef struct gr4hip_fir_interp gr4hip_fir_interp_t;
int gr4hip_fir_interp_create(gr4hip_fir_interp_t** fir, int dtype, const float* h_taps, size_t ntaps, size_t interp);
int gr4hip_fir_interp_set_taps(gr4hip_fir_interp_t* fir, const float* h_taps, size_t ntaps); /* history kept unless it must grow (like fir_filter) */
int gr4hip_fir_interp_reset(gr4hip_fir_interp_t* fir);
int gr4hip_fir_interp_process(gr4hip_fir_interp_t* fir, const void* d_in, size_t n_in, void* d_out, size_t* n_out, gr4hip_stream_t stream);
int gr4hip_fir_interp_destroy(gr4hip_fir_interp_t* fir);

/* gr::filter::Decimator<T>::processBulk (time_domain_filter.hpp:234-244): keep samples with i % decim == 0. */
int gr4hip_decimate(int dtype, const void* d_in, size_t n_in, size_t decim, void* d_out, size_t* n_out, gr4hip_stream_t stream);

/* ------------------------------------------------------------------------------------------------ a3/a4
 * gr::filter::Filter<float>::processOne == cascade of sections through detail::computeFilter
 * (algorithm/.../filter/FilterTool.hpp:116-158, 244-246), and gr::filter::iir_filter<float, form>::processOne
 * (time_domain_filter.hpp:89-121) for nsections == 1.  b: [nsections][nb], a: [nsections][na], a[.][0] == 1.
 * All four forms compute the same transfer function from zero state and are EVALUATED as direct form II (the parallel-in-time scan works on the DF-II
 * state); `form` is recorded for introspection only, so rounding can differ from the reference's DF_I / transposed forms in the last bits (the four forms
 * agree to 1e-5 upstream too, qa_filter.cpp:53-128).
 * Evaluation: when the cascade's memory fades within 1, 2 or 4 tiles of 8192 samples (||Phi^tiles||_inf <= 1e-8, checked in float64 at create), a span is cut
 * into contiguous runs of tiles that start from a warm-up over the preceding tiles instead of the exact state: the state a run starts from is within 1e-8
 * (relative to the state that far back) of the exact one -- three orders below the float32 parity tolerance; the first run of every call starts from the
 * handle's carried state exactly.  Filters that fade more slowly (poles within ~5e-4 of the unit circle) take the exact look-back scan.
 * gr4hip_iir_status: synchronises `stream` and reports a look-back time-out of an earlier launch of this handle (a bounded wait gave up: never observed,
 * but then that call's output is invalid) as GR4HIP_RUNTIME_ERROR; the next process / reset call reports it too. */
typedef struct gr4hip_iir gr4hip_iir_t;
int gr4hip_iir_create(gr4hip_iir_t** iir, int form, size_t nsections, const float* h_b, size_t nb, const float* h_a, size_t na);
int gr4hip_iir_reset(gr4hip_iir_t* iir);
int gr4hip_iir_process(gr4hip_iir_t* iir, const float* d_in, size_t n, float* d_out, gr4hip_stream_t stream);
int gr4hip_iir_status(gr4hip_iir_t* iir, gr4hip_stream_t stream);
/* How the cascade is evaluated (per handle; changing it restarts the filter from zero state):
 *   GR4HIP_IIR_PARALLEL        the parallel-in-time kernels described above (direct form II whatever `form` says).
 *   GR4HIP_IIR_SEQUENTIAL_F32  the reference's own arithmetic: one lane walks the cascade sample by sample in the REQUESTED form (detail::computeFilter,
 *                              FilterTool.hpp:116-158: DF_I / DF_II / DF_I_TRANSPOSED / DF_II_TRANSPOSED with their own input / output histories), float32
 *                              operations in source order.  ~10 Msamples/s; as close to float64 as the block on the host, form for form.
 *   GR4HIP_IIR_AUTO (default)  PARALLEL unless float32 cannot carry this cascade's state through the scan: gr4hip_iir_create measures it -- a three-tile noise vector
 *                              through the kernels the handle would use, against float64 and against the sequential float32 form on the host -- and selects
 *                              SEQUENTIAL_F32 when the parallel result is off by more than 1e-5 of the output rms AND by more than ten times the sequential float32
 *                              error (ill-conditioned narrow-band cascades of high order: an order-16, fc = 0.016 Butterworth measured 7.8e-2 against 7.0e-3).
 * gr4hip_iir_get_algo reports the evaluation in use and the two create-time errors (max |error| / output rms; < 0: not measured). */
typedef enum { GR4HIP_IIR_AUTO = 0, GR4HIP_IIR_PARALLEL = 1, GR4HIP_IIR_SEQUENTIAL_F32 = 2 } gr4hip_iir_algo;
int gr4hip_iir_set_algo(gr4hip_iir_t* iir, int algo);
int gr4hip_iir_get_algo(const gr4hip_iir_t* iir, int* algo_in_use, float* selftest_parallel, float* selftest_sequential_f32);
int gr4hip_iir_destroy(gr4hip_iir_t* iir);

/* BasicDecimatingFilter (FIR design) -> BasicFilter (IIR design), BASELINE.json configs[2]: the decimating processBulk loop (time_domain_filter.hpp:190-204) feeding
 * Filter<float>::processOne over the sections (FilterTool.hpp:244-246).  One call for both handles; each keeps its own state (history, delay lines) exactly as if
 * gr4hip_fir_process and gr4hip_iir_process were called one after the other, and the two may be mixed freely with those calls.
 * mode GR4HIP_FIR_IIR_ONE_LAUNCH: when the filter takes the frequency-domain decimator (float, decimate by 8, <= 1024 taps, a span of >= 64 blocks of 7168 samples) and
 * the cascade is up to four biquads on the parallel path whose memory fades within four blocks, the cascade runs as the decimator's STORE EPILOGUE: one launch, the
 * decimated stream never reaches HBM (4.5 instead of 5.5 bytes per input sample) -- contiguous block runs per workgroup, the cascade's state carried in one wave,
 * every run but the first warmed up over the blocks in front of it (state error <= 1e-8); anything else runs as the two launches.
 * mode GR4HIP_FIR_IIR_TWO_LAUNCHES: the decimated stream through a scratch buffer of the filter handle.
 * mode GR4HIP_FIR_IIR_AUTO: the faster of the two as measured on MI355X -- the two launches (the decimator is bound by vector-instruction issue, not by HBM: the
 * cascade's instructions cost the same inside its launch as in their own, and the contiguous runs cost the decimator another 10 %; DESIGN.md 7 has the numbers).
 * ONE_LAUNCH is for a caller that shares HBM bandwidth with other streams and prefers the smaller traffic.
 * The decimator's dynamic-range guard works as in gr4hip_fir_process (a rejected span is redone on the polyphase kernels + the cascade's own, from untouched states). */
typedef enum { GR4HIP_FIR_IIR_AUTO = 0, GR4HIP_FIR_IIR_ONE_LAUNCH = 1, GR4HIP_FIR_IIR_TWO_LAUNCHES = 2 } gr4hip_fir_iir_mode;
int gr4hip_fir_iir_process(gr4hip_fir_t* fir, gr4hip_iir_t* iir, const float* d_in, size_t n_in, float* d_out, size_t* n_out, int mode, gr4hip_stream_t stream);

/* ------------------------------------------------------------------------------------------------ a5 (host-side design)
 * BasicFilterProto<float>::designFilter (time_domain_filter.hpp:163-182): FIR by window method
 * (fir::designFilter<float>, FilterTool.hpp:964-1071; tap count from Kaiser's estimate :985-1004) or IIR biquad
 * sections (iir::designFilter<float>, :476-917).  Pure host code; results feed gr4hip_fir_create / gr4hip_iir_create. */
typedef struct { /* gr::filter::FilterParameters (FilterTool.hpp:66-75) */
    size_t order;
    double f_low, f_high, gain, ripple_db, attenuation_db, beta, fs;
} gr4hip_filter_params;
typedef enum { GR4HIP_LOWPASS = 0, GR4HIP_HIGHPASS, GR4HIP_BANDPASS, GR4HIP_BANDSTOP } gr4hip_filter_response; /* filter::Type :64 */
typedef enum { GR4HIP_BUTTERWORTH = 0, GR4HIP_BESSEL, GR4HIP_CHEBYSHEV1, GR4HIP_CHEBYSHEV2 } gr4hip_iir_design_t; /* iir::Design :425-430 */
int gr4hip_filter_params_default(gr4hip_filter_params* p);
/* h_taps == NULL: only *ntaps is returned (size query) */
int gr4hip_fir_design(int response, const gr4hip_filter_params* p, int window, float* h_taps, size_t cap, size_t* ntaps);
/* biquads: h_b, h_a are [cap_sections][3] (first-order sections are zero padded) */
int gr4hip_iir_design(int response, const gr4hip_filter_params* p, int design, float* h_b, float* h_a, size_t cap_sections, size_t* nsections);

/* ------------------------------------------------------------------------------------------------ a7-a10
 * gr::blocks::fft::FFT<T>::processBulk (blocks/fourier/.../fft.hpp:147-171) per frame of fft_size samples:
 * window (window.hpp:69-183, default Hann) -> forward DFT (algorithm/.../fourier/fft.hpp:113-153) ->
 * magnitude hypot*2/N [dB] fft-shifted + phase atan2 [unwrap][deg] fft-shifted (fft_common.hpp:20-123) + Re + Im
 * (natural order, fft.hpp:217-220).  in_dtype C32: fft_size values per output; F32: fft_size/2 (fft.hpp:140-143, 221-227).
 * Any output pointer may be NULL.  d_ranges (optional): per frame {min,max} of mag, phase, re, im (fft.hpp:229-232).
 * fft_size: any power of two up to 2^20 (one kernel up to 8192, a four-step pipeline above) and every other size up to 2^19 -- the
 * sizes SimdFFT takes with radix-3/5 passes (SimdFFT.hpp:348-375) and the ones the reference sends to Bluestein
 * (algorithm/.../fourier/fft.hpp:353-381) alike run as a chirp convolution over the power-of-two transforms; larger sizes return
 * GR4HIP_UNSUPPORTED from create (the caller keeps its CPU path). */
typedef struct gr4hip_fft gr4hip_fft_t;
int gr4hip_fft_create(gr4hip_fft_t** fft, int in_dtype, size_t fft_size, int window, int flags);
int gr4hip_fft_process(gr4hip_fft_t* fft, const void* d_in, size_t n_frames, float* d_mag, float* d_phase, float* d_re, float* d_im,
                       float* d_ranges, gr4hip_stream_t stream);
/* raw spectrum of algorithm::FFT::compute (interleaved complex, natural order, window applied if configured) */
int gr4hip_fft_spectrum(gr4hip_fft_t* fft, const void* d_in, size_t n_frames, float* d_spectrum, gr4hip_stream_t stream);
/* |X[k]|^2 in natural bin order: mag2[(k + N/2) % N] == (magnitude_block[k] * N/2)^2 */
int gr4hip_fft_mag2(gr4hip_fft_t* fft, const void* d_in, size_t n_frames, float* d_mag2, gr4hip_stream_t stream);
/* Per-sample float blocks BEHIND a power spectrum (MultiplyConst / DivideConst / AddConst / SubtractConst<float> on the |X|^2 stream: normalisation, an offset) in the
 * transform's launch: the program (gr4hip_ewise_t of dtype F32, copied; NULL or empty removes it) is applied to every |X|^2 value of gr4hip_fft_mag2 before its store --
 * natively in every transform kernel of the FFT block (one launch); the 8192-point frame pipeline runs it as one element-wise launch over its output inside the call. */
int gr4hip_fft_set_epilogue(gr4hip_fft_t* fft, const gr4hip_ewise_t* f32_prog);
int gr4hip_fft_destroy(gr4hip_fft_t* fft);
/* Which device path a size takes (host-only, no device needed): kind 0 = power of two <= 8192 (one kernel), 3 = {2,3,5}-smooth <= 8192 (mixed-radix passes in one
 * launch: the sizes SimdFFT::canProcessSize takes with radix-3 / radix-5 passes, SimdFFT.hpp:348-375; radices[0 .. n_passes) is the run-time plan, <= 15 passes of
 * radix 2 .. 16), 1 = power of two up to 2^20 (four-step), 2 = any other size up to 2^19 (chirp convolution); GR4HIP_UNSUPPORTED beyond. */
int gr4hip_fft_plan(size_t fft_size, int* kind, int* radices16, int* n_passes);
/* window::create on the host (float), for callers that need the block's _window member */
int gr4hip_window_create(int window, float* h_out, size_t n, float beta);
int gr4hip_window_create_f64(int window, double* h_out, size_t n, double beta); /* create<double> */

/* ------------------------------------------------------------------------------------------------ headline chain
 * complex<float> fir_filter -> FFT block frames -> |X|^2 (BASELINE.json configs[1]).  One call consumes
 * floor(n_samples / fft_size) frames; FIR history (ntaps-1 samples) is carried across calls.
 * The runtime analogue of Merge<fir,"out",fft,"in"> (core/include/gnuradio-4.0/BlockMerging.hpp:136-320). */
typedef struct gr4hip_chain gr4hip_chain_t;
int gr4hip_chain_create(gr4hip_chain_t** chain, const float* h_taps, size_t ntaps, size_t fft_size, int window, int algo);
int gr4hip_chain_reset(gr4hip_chain_t* chain);
int gr4hip_chain_process(gr4hip_chain_t* chain, const void* d_in_c32, size_t n_samples, float* d_mag2, size_t* n_frames,
                         gr4hip_stream_t stream);
int gr4hip_chain_get_algo(const gr4hip_chain_t* chain, int* algo_in_use);
/* Dynamic-range guard of GR4HIP_CHAIN_AUTO.  The fused kernels filter in the frequency domain and carry the float32 rounding of their transforms, which is sized by the
 * frame's INPUT, while the parity bar (1e-5 of max(|Y_k|^2, rms_k |Y_k|^2)) is sized by its OUTPUT.  Every fused launch of an AUTO chain therefore judges EVERY frame (all of its
 * samples and bins) on two statistics (round 6; chain_fused.hip kGuardR4Max / kGuardPeakMax have the measurements):
 *   R4 = w2 nf mean|x|^2 / rms_k |Y_k|^2          the spread error, K sqrt(R4) with K <= 1.1e-6 on noise-like input: wide-band input the filter removes most of (a 1 %-pass-band
 *                                                filter over white noise has R4 = 6 .. 13, a 4 % one 3 .. 5, a wide-band neighbour 20 dB up ~50);
 *   T' = 2 wg^2 peak(X)^2 / rms_k |Y_k|^2         the images a float32 transform leaves of a strong line (1.2 eps of it, half a transform away): a line the filter takes down by
 *                                                10 dB or more that still dominates the output, error <= 1.4e-7 sqrt(T');
 * a frame is marked when R4 / 20 + T' / 2000 > 1 (the errors add in power: <= 6.3e-6 on the boundary; fftSize < 8192, judged per 8192-sample block: R4 / 8 + T' / 600).
 * (Until round 6 the test was output / input power < 0.08, which marked every frame of every narrow filter where nothing needed fixing, and missed the second case.)
 * What happens with the verdicts is the handle's guard mode (gr4hip_chain_set_guard_mode below; default STRICT: the frames a launch marks are evaluated again in the time domain
 * by launches enqueued behind it -- on the f16 matrix pipe where that agrees with the fused result, in float64 where it does not -- and a stream in which more than a tenth of
 * all frames end in float64 moves to the direct-form kernels, history handed over, where it stays until gr4hip_chain_reset).  Explicit GR4HIP_CHAIN_FUSED_FD never measures nor
 * switches.  gr4hip_chain_last_power_ratio waits for the last measured launch and returns its output / input power (window gain taken out; < 0: nothing measured yet --
 * informative since round 6) and whether the chain now runs in the time domain; gr4hip_chain_last_guard_fractions the fraction of that launch's frames the kernel marked
 * (< 0: nothing measured yet) and the fraction of all frames since create / reset that went on to float64. */
int gr4hip_chain_last_power_ratio(gr4hip_chain_t* chain, float* ratio, int* time_domain, gr4hip_stream_t stream);
int gr4hip_chain_last_guard_fractions(gr4hip_chain_t* chain, float* marked, float* float64, gr4hip_stream_t stream);
/* What the guard does with its measurement (per handle; GR4HIP_CHAIN_AUTO chains on the fused frequency-domain kernel only, a no-op elsewhere):
 *   GR4HIP_GUARD_STRICT (default): nothing out of tolerance is ever handed out, and nobody waits.  The fused kernel writes one verdict byte per frame; chain_td16_kernel
 *     (8192-point frames: the filter with 22-bit products on the f16 matrix pipe, window, frame transform; stored where it agrees with the fused result in every bin)
 *     and chain_redo_kernel (what that leaves: y = sum b[k] x[n - k] on the FP64 matrix pipe, window, frame transform, |.|^2 over the fused result), enqueued behind it
 *     on the same stream, evaluate exactly the marked frames again (an ordinary stream costs two near-empty launches, ~5 us each).  gr4hip_chain_process returns when
 *     the launches are queued ("user code must not block in work()", docs/USER_API_advanced_work.md).  The counts of EARLIER launches, once they have arrived, move a
 *     stream in which more than a tenth of all frames ended in float64 to the direct-form kernels for good -- read without waiting.  Several chains in one call
 *     (gr4hip_chain_process_multi) work the same way: per-channel verdict bytes and one chain_redo_kernel per channel, or -- the fold -- a frame marked when ANY
 *     channel's share of it is and chain_redo_fold_kernel, which evaluates every channel of a marked frame again and keeps the sum in registers.
 *   GR4HIP_GUARD_DEFERRED: calls stay asynchronous.  The first call after create / reset probes its first 8 blocks synchronously; later calls read the finished
 *     measurements of EARLIER launches, so the call in which a strong out-of-band signal first appears is published from the fused kernel (error floor
 *     ~2e-6 of the input rms) and the switch happens from the next call on.
 *   GR4HIP_GUARD_OFF: no measurement, never switches (what an explicit GR4HIP_CHAIN_FUSED_FD chain does). */
int gr4hip_chain_set_guard_mode(gr4hip_chain_t* chain, int mode);
/* n_chains (<= 16) chains of the same fft size fed n_samples each in ONE call -- the branches of a flowgraph with parallel SDR channels that share a device
 * (BASELINE.json configs[4] at fewer GPUs than channels).  d_in_c32 / d_mag2 are HOST arrays of device pointers.
 *   d_mag2 != NULL: chain i's spectra go to d_mag2[i];  d_sum != NULL: sum_i |FFT(fir(x_i))|^2, the combiner MathOpMultiPortImpl<float, std::plus>
 *   (blocks/math/.../Math.hpp:73-108, left fold over the inputs) -- both may be given.
 * When every chain runs the fused frequency-domain kernel at 8192 points with the rectangular window, the call is ONE persistent launch (one workgroup per
 * CU) instead of n kernels contending for every CU; with d_mag2 == NULL and identical taps on all chains the fold is kept in registers and only d_sum is
 * written (8 + 4 / n bytes of HBM traffic per sample).  Any other combination is served chain by chain followed by gr4hip_math_nary, with the same results.
 * History, guard state and measurements stay per chain (with the in-register fold the guard measures all channels together: the ratio that matters for
 * the delivered sum).  Do not mix this call with calls on the same handles from other streams. */
int gr4hip_chain_process_multi(gr4hip_chain_t* const* chains, size_t n_chains, const void* const* d_in_c32, size_t n_samples, float* const* d_mag2,
                               float* d_sum, size_t* n_frames, gr4hip_stream_t stream);
/* The fused kernels are persistent: one workgroup per CU that takes ALL of the CU's registers and LDS, so nothing else (e.g. the RCCL
 * kernels of a fan-in collective on another stream) runs beside them.  n > 0 caps the grid at n workgroups and leaves the other CUs free;
 * 0 = all CUs (default).  No effect on the unfused path. */
int gr4hip_chain_set_max_workgroups(gr4hip_chain_t* chain, unsigned n);
int gr4hip_chain_destroy(gr4hip_chain_t* chain);

/* ------------------------------------------------------------------------------------------------ e: the cross-device combiner edge
 * A flowgraph shards across the GPUs of a node only along independent branches (one SDR channel per GPU); the one exchange step is the
 * combiner MathOpMultiPortImpl<float, std::plus> (blocks/math/.../Math.hpp:73-108) whose inputs sit on different "gpu:hip:i" compute domains
 * (ComputeDomain.hpp:47-100; EdgeParameters.domain, BlockModel.hpp:64-72).  Every rank folds its own channels (gr4hip_chain_process_multi /
 * gr4hip_math_nary) and these calls add the partial sums over RCCL (xGMI): one process per GPU, one communicator rank per process, the collective
 * queued on a caller stream like every other call here.  librccl is opened at run time (an RCCL the process already carries is reused; GR4HIP_RCCL_LIBRARY
 * names one explicitly) -- GR4HIP_UNSUPPORTED with the reason in gr4hip_last_error() when there is none.
 *   gr4hip_fanin_unique_id   rank 0 draws the 128-byte id; the caller ships it to the other ranks (file, socket, MPI, a torch.distributed store ...)
 *   gr4hip_fanin_create      collective over all ranks; binds the communicator to the calling thread's current device (gr4hip_set_device)
 *   ..._reduce_scatter_sum   d_partial: n_ranks shards of shard_count floats (this rank's partial sum of every shard); d_shard: the all-rank sum of shard `rank`
 *   ..._all_to_all_sum       the same result as point-to-point sends (one xGMI link per peer, all busy at once) + a left fold in RANK order: the reduction
 *                            order does not depend on RCCL's ring; d_scratch: n_ranks * shard_count floats
 *   ..._all_reduce_sum       every rank gets the whole sum (a sink that lives on every rank, or on one)
 * In-place operation (d_shard inside d_partial etc.) follows RCCL's rules. */
typedef struct gr4hip_fanin gr4hip_fanin_t;
#define GR4HIP_FANIN_ID_BYTES 128
int gr4hip_fanin_unique_id(void* id128);
int gr4hip_fanin_create(gr4hip_fanin_t** fanin, const void* id128, int rank, int n_ranks);
int gr4hip_fanin_rank(const gr4hip_fanin_t* fanin, int* rank, int* n_ranks);
int gr4hip_fanin_reduce_scatter_sum_f32(gr4hip_fanin_t* fanin, const float* d_partial, float* d_shard, size_t shard_count, gr4hip_stream_t stream);
int gr4hip_fanin_all_to_all_sum_f32(gr4hip_fanin_t* fanin, const float* d_partial, float* d_scratch, float* d_shard, size_t shard_count, gr4hip_stream_t stream);
int gr4hip_fanin_all_reduce_sum_f32(gr4hip_fanin_t* fanin, const float* d_partial, float* d_sum, size_t count, gr4hip_stream_t stream);
int gr4hip_fanin_destroy(gr4hip_fanin_t* fanin);

/* ------------------------------------------------------------------------------------------------ a11/a12/a13
 * MathOpImpl<T,op>::processOne (blocks/math/.../Math.hpp:38-56): out = in (op) value, C++ semantics for T
 * (integer promotion then narrowing, wrap-around).  h_value points to one host element of `dtype`.
 * Spans need only their element's natural alignment (a ring span starts at any element); 16-byte aligned spans are fastest. */
int gr4hip_math_const(int op, int dtype, const void* d_in, void* d_out, size_t n, const void* h_value, gr4hip_stream_t stream);
/* MathOpMultiPortImpl<T,op>::processBulk (Math.hpp:100-107): left fold ((in0 op in1) op in2) ... over 1..32 inputs.
 * h_d_ins is a HOST array of n_inputs device pointers. */
int gr4hip_math_nary(int op, int dtype, const void* const* h_d_ins, size_t n_inputs, void* d_out, size_t n, gr4hip_stream_t stream);

/* ------------------------------------------------------------------------------------------------ fused runs of per-sample blocks
 * The reference fuses adjacent blocks at compile time: Merge<A, "out", B, "in"> (core/include/gnuradio-4.0/BlockMerging.hpp:126-240) is ONE processOne() that
 * carries the value through both parts in registers -- its published table (docs/USER_API_Connecting_Blocks.md:207-222) measures mult -> div -> add and that
 * chain ten times over.  gr4hip_ewise is the same thing for chains known at run time: a PROGRAM of per-sample ops of one sample type --
 *   gr4hip_ewise_append_const    MathOpImpl<T, op>::processOne (Math.hpp:38-56): value (op) constant, the semantics of gr4hip_math_const (integers promote, wrap and
 *                                narrow like C++: bit-exact; float ops are single IEEE operations in program order, never contracted into fused multiply-adds)
 *   gr4hip_ewise_append_rotator  Rotator<complex<float>>::processOne (Rotator.hpp:51-61) with the closed-form phase of GR4HIP_ROTATOR_CLOSED_FORM: sample k of the
 *                                stream (counted from create / reset) is multiplied by exp(j (initial_phase + (k + 1) phase_increment)); C32 programs only
 * -- that gr4hip_ewise_process runs as ONE launch with the values in registers: 2 sizeof(T) bytes of HBM traffic per sample whatever the number of ops.
 * The same program can ride in the launch of a neighbouring filter as its load hook (prologue) or store hook (epilogue): gr4hip_fir_set_prologue / _epilogue below.
 * A program holds no device state besides its op list; the stream position (what a rotator op's phase is a function of) is advanced by gr4hip_ewise_process. */
int gr4hip_ewise_create(gr4hip_ewise_t** prog, int dtype);
int gr4hip_ewise_append_const(gr4hip_ewise_t* prog, int op, const void* h_value); /* h_value: one host element of the program's dtype */
int gr4hip_ewise_append_rotator(gr4hip_ewise_t* prog, float phase_increment, float initial_phase);
int gr4hip_ewise_length(const gr4hip_ewise_t* prog, size_t* n_ops);
int gr4hip_ewise_reset(gr4hip_ewise_t* prog); /* stream position back to 0: rotator ops restart from their initial phase */
int gr4hip_ewise_position(const gr4hip_ewise_t* prog, uint64_t* samples);
int gr4hip_ewise_process(gr4hip_ewise_t* prog, const void* d_in, void* d_out, size_t n, gr4hip_stream_t stream); /* in place (d_in == d_out) is allowed */
/* gr::filter::Decimator<T> (time_domain_filter.hpp:234-244) with the program's blocks BEHIND it in one launch: d_out[m] = program(d_in[m * decim]), m < ceil(n_in / decim);
 * only the kept samples are read.  The stream position (rotator phase) counts OUTPUT samples.  Memoryless blocks in front of a Decimator commute with it, so a planner
 * that finds const blocks on either side of one hands them all to this call. */
int gr4hip_ewise_decimate(gr4hip_ewise_t* prog, const void* d_in, size_t n_in, size_t decim, void* d_out, size_t* n_out, gr4hip_stream_t stream);
int gr4hip_ewise_destroy(gr4hip_ewise_t* prog);
/* Neighbours of a FIR filter in ITS launch.  prologue: applied to every input sample before the filter sees it (the history the filter carries is the history of
 * the prologue's OUTPUT, zero before the first sample, exactly as if the blocks ran one after the other); epilogue: applied to every output sample before it is
 * stored.  The program is copied (later changes to `prog` do not reach the filter); NULL or an empty program removes the hook.  dtype of the program == dtype of
 * the filter.  How it is executed:
 *   - a program that is nothing but real gains (MultiplyConst / DivideConst; complex constants with a zero imaginary part) is folded into the taps -- a FIR filter
 *     is linear, fir(g x) == (g b) * x -- and costs nothing in any kernel (the rounding differs from the two-block form in the last bits: one float product per
 *     tap instead of one per sample, same float32 level, same 1e-5 parity bar);
 *   - anything else (AddConst / SubtractConst, complex gains, a rotator) is a load / store hook of the register-window kernel (any tap count, any decimation, float
 *     or complex) or of the band-form matrix-pipe decimators (float: decimation 2 .. 12, complex: 3 .. 16, windows they hold, spans of >= 2^14 outputs): one launch, no
 *     intermediate stream in HBM.  Exception, by measurement: where a plain filter of the same shape takes a matrix-pipe or
 *     frequency-domain kernel AND the register-window kernel is far behind it (more than 96 taps -- 64 for complex -- on a span of >= 2^16 samples, float decimators with
 *     more than 12 taps per output on long spans), that kernel is worth more than the saved pass
 *     (add -> 256-tap FIR: 165 Gsamples/s hooked, 277 as an element-wise launch + the bf16 kernel), and the program runs as ONE element-wise launch in front of
 *     (behind) the filter's own inside this call: same results, same carried history.
 * The history the filter carries is always what the prologue in force made of the samples -- also when the prologue is replaced in mid-stream: the samples already in
 * the filter's memory keep the OLD program's values, exactly as when the blocks run one after the other.  GR4HIP_UNSUPPORTED: the program's dtype does not match. */
int gr4hip_fir_set_prologue(gr4hip_fir_t* fir, const gr4hip_ewise_t* prog);
int gr4hip_fir_set_epilogue(gr4hip_fir_t* fir, const gr4hip_ewise_t* prog);

/* Rotator<complex<float>>::processOne (blocks/math/.../Rotator.hpp:51-61): phase += inc (before the first sample),
 * single +-2pi wrap, y = x * (cos, sin); the accumulated phase is carried across calls.
 * Two evaluations of the phase (gr4hip_rotator_set_algo; both keep the same carried state, so they may be switched between calls):
 *   GR4HIP_ROTATOR_CLOSED_FORM (default): sample i of a call sees carried + (i + 1) inc, computed in float64 and reduced exactly -- the float64
 *     oracle's phase (<= 1e-5 against it however long the stream and however many calls it is cut into: the carried phase is kept in float64 by the
 *     handle, gr4hip_rotator_phase reports it rounded to float), one HBM-bound pass.  The default is MORE ACCURATE THAN, not identical to, the
 *     reference block: it does NOT reproduce the drift of the reference's float accumulator (~1e-7 rad per step), so against the reference's own output
 *     it is beyond 1e-5 after ~10^3..10^5 samples.  A caller who needs the CPU block's output bit for bit selects the recurrence (below; the host
 *     engine: gr::hip::options().rotator_reference_recurrence).
 *   GR4HIP_ROTATOR_RECURRENCE: the reference's float recurrence itself, bit-identical phase sequence and carried phase (a sequential walk: ~10^2..10^4
 *     times slower, for callers that need the reference's exact output). */
typedef struct gr4hip_rotator gr4hip_rotator_t;
typedef enum { GR4HIP_ROTATOR_CLOSED_FORM = 0, GR4HIP_ROTATOR_RECURRENCE = 1 } gr4hip_rotator_algo;
int gr4hip_rotator_create(gr4hip_rotator_t** rot, float phase_increment, float initial_phase);
int gr4hip_rotator_set_algo(gr4hip_rotator_t* rot, int algo);
int gr4hip_rotator_reset(gr4hip_rotator_t* rot, float initial_phase); /* settingsChanged: _accumulated_phase = initial_phase */
int gr4hip_rotator_process(gr4hip_rotator_t* rot, const void* d_in_c32, void* d_out_c32, size_t n, gr4hip_stream_t stream);
int gr4hip_rotator_phase(gr4hip_rotator_t* rot, float* phase, gr4hip_stream_t stream); /* synchronises the stream */
int gr4hip_rotator_destroy(gr4hip_rotator_t* rot);

/* ------------------------------------------------------------------------------------------------ float64 instantiations
 * The second registered type of the hot-path blocks: fir_filter<double> / iir_filter<double, form> (time_domain_filter.hpp:20, 57-60), FFT<double>
 * (fourier/fft.hpp:29: real double frames -> DataSet<double>) and Rotator<complex<double>> (Rotator.hpp:15).  Same semantics as the float32 entry points
 * above (history / state carried between calls, decimation = y[m D], all IIR forms evaluated as DF-II, FFT outputs of a real-input block: magnitude and
 * phase of bins 0 .. N/2-1, Re / Im of bins N/2 .. N-1), plain FP64 kernels.  Envelope: FIR <= 2048 taps, decim <= 32; IIR <= 8 state values (4 biquads,
 * or one section of order <= 8); FFT powers of two 2 .. 8192; anything else returns GR4HIP_UNSUPPORTED from create (the caller keeps its CPU path). */
typedef struct gr4hip_fir64 gr4hip_fir64_t;
int gr4hip_fir64_create(gr4hip_fir64_t** fir, const double* h_taps, size_t ntaps, size_t decim);
int gr4hip_fir64_set_taps(gr4hip_fir64_t* fir, const double* h_taps, size_t ntaps); /* history kept unless it has to grow */
int gr4hip_fir64_reset(gr4hip_fir64_t* fir);
int gr4hip_fir64_process(gr4hip_fir64_t* fir, const double* d_in, size_t n_in, double* d_out, size_t* n_out, gr4hip_stream_t stream);
int gr4hip_fir64_destroy(gr4hip_fir64_t* fir);
typedef struct gr4hip_iir64 gr4hip_iir64_t;
int gr4hip_iir64_create(gr4hip_iir64_t** iir, int form, size_t nsections, const double* h_b, size_t nb, const double* h_a, size_t na);
int gr4hip_iir64_reset(gr4hip_iir64_t* iir);
int gr4hip_iir64_process(gr4hip_iir64_t* iir, const double* d_in, size_t n, double* d_out, gr4hip_stream_t stream);
int gr4hip_iir64_destroy(gr4hip_iir64_t* iir);
typedef struct gr4hip_fft64 gr4hip_fft64_t;
int gr4hip_fft64_create(gr4hip_fft64_t** fft, size_t fft_size, int window, int flags);
int gr4hip_fft64_process(gr4hip_fft64_t* fft, const double* d_in, size_t n_frames, double* d_mag, double* d_phase, double* d_re, double* d_im, gr4hip_stream_t stream);
int gr4hip_fft64_destroy(gr4hip_fft64_t* fft);
typedef struct gr4hip_rotator64 gr4hip_rotator64_t;
int gr4hip_rotator64_create(gr4hip_rotator64_t** rot, double phase_increment, double initial_phase);
int gr4hip_rotator64_reset(gr4hip_rotator64_t* rot, double initial_phase);
int gr4hip_rotator64_process(gr4hip_rotator64_t* rot, const void* d_in_c64, void* d_out_c64, size_t n, gr4hip_stream_t stream);
int gr4hip_rotator64_phase(gr4hip_rotator64_t* rot, double* phase);
int gr4hip_rotator64_destroy(gr4hip_rotator64_t* rot);

/* ------------------------------------------------------------------------------------------------ batched FIR (configs[3])
 * nchannels independent fir_filter<float> instances, per-channel taps h_taps[c][k], channel-major samples
 * d_in[c * in_stride + n]; evaluated as a block-Toeplitz contraction on the matrix pipe: more than 32 taps on spans of >= 32768 samples per channel with two-term
 * f16 splits under a per-segment block exponent (csrc/fir_f16.hip: three products per tap, every segment judged, see gr4hip_fir_set_algo above), otherwise with
 * float32 products on the f32 MFMA units. */
typedef struct gr4hip_fir_batched gr4hip_fir_batched_t;
int gr4hip_fir_batched_create(gr4hip_fir_batched_t** fb, size_t nchannels, const float* h_taps, size_t ntaps);
int gr4hip_fir_batched_reset(gr4hip_fir_batched_t* fb);
int gr4hip_fir_batched_process(gr4hip_fir_batched_t* fb, const float* d_in, size_t in_stride, size_t n, float* d_out, size_t out_stride,
                               gr4hip_stream_t stream);
int gr4hip_fir_batched_destroy(gr4hip_fir_batched_t* fb);

/* ------------------------------------------------------------------------------------------------ a15 (bench input)
 * device-side synthetic stream of SURVEY.md 8(d): unit-power Gaussian noise + tone.  The stream is cut into groups of 8 samples; group g has its own
 * generator Xoshiro256pp(seed ^ 0xd1b54a32d192ed03 (g + 1)) (algorithm/.../rng/Xoshiro256pp.hpp:22-96, constructor :33-39) and fills its 8 samples with
 * GaussianNoise<float>::fill / fillComplex (GaussianNoise.hpp:59-111: Marsaglia polar, amplitude / sqrt2 per component for complex), so every group
 * is the reference's own recipe and the groups are independent (any lane can generate any group).  It is NOT the single sequential reference stream
 * (tests that need that one upload it).  gr4hip_synth_draws returns the raw 64-bit draws of group g's generator: with seed = 0xd1b54a32d192ed03, g = 0
 * that generator is Xoshiro256pp(0), whose first draws are the reference's known answer (algorithm/test/qa_Xoshiro256pp.cpp:55-69). */
int gr4hip_synth_draws(uint64_t* d_out_u64, size_t n_draws, uint64_t seed, uint64_t group, gr4hip_stream_t stream);
int gr4hip_synth_c32(void* d_out_c32, size_t n, uint64_t seed, double tone_frel, float tone_amp, float noise_amp, gr4hip_stream_t stream);
int gr4hip_synth_f32(float* d_out, size_t n, uint64_t seed, double tone_frel, float tone_amp, float noise_amp, gr4hip_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* GR4HIP_H */
