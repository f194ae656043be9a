// chain.hip -- the headline chain: complex<float> fir_filter -> FFT block frames -> |X|^2 (BASELINE.json configs[1]).
//
// GR4HIP_CHAIN_UNFUSED: gr4hip_fir_process -> y in HBM -> gr4hip_fft_mag2 (any size the FFT block supports).
// GR4HIP_CHAIN_FUSED_FD (chain_fused.hip): one persistent launch, any window, fft_size 256...8192, <= 256 taps; AUTO picks it when it applies.
// GR4HIP_CHAIN_FUSED_TD (chain_td.hip): one launch, direct-form filter on the matrix pipe + one transform per frame, fft_size 256...4096, <= 256 taps;
//   AUTO picks it for <= 64 taps (faster than the fast convolution there).  The dynamic-range guard sends a stream to the kernel pair with float32 products.
#include "common.hpp"
#include <cstdlib>

namespace gr4 {
int chain_fused_supported(size_t ntaps, size_t fft_size, int window, int algo);
struct ChainFused;
int  chain_fused_create(ChainFused** out, const float* taps, size_t ntaps, size_t fft_size, int window, int algo);
int  chain_fused_reset(ChainFused* c);
int  chain_fused_process(ChainFused* c, const float* d_in, size_t n_frames, float* d_mag2, hipStream_t st);
int  chain_fused_process_multi(ChainFused* const* cs, size_t n, bool shared_taps, const float* const* d_in, size_t n_frames, float* const* d_out, float* d_sum, hipStream_t st, bool redo);
bool chain_fused_multi_capable(const ChainFused* c);
void chain_fused_destroy(ChainFused* c);
void chain_fused_set_max_workgroups(ChainFused* c, unsigned n);
void chain_fused_set_measure(ChainFused* c, bool on);
void chain_fused_set_redo(ChainFused* c, bool on); // measured launches mark their frames one by one and chain_redo_kernel follows them
int  chain_fused_power_ratio(ChainFused* c, bool wait, bool fir_output, float* ratio, float* marked_fraction = nullptr, float* float64_fraction = nullptr);
const float* chain_fused_history(ChainFused* c, hipStream_t st); // (applies a pending reset on `st` first; null: that failed)
int  chain_fused_set_history(ChainFused* c, const float* d_hist256, hipStream_t st);
struct ChainTd;
int  chain_td_supported(size_t ntaps, size_t fft_size, int window);
int  chain_td_create(ChainTd** out, const float* taps, size_t ntaps, size_t fft_size, int window);
int  chain_td_reset(ChainTd* c);
int  chain_td_process(ChainTd* c, const float* d_in, size_t n_frames, float* d_mag2, hipStream_t st, bool judged);
const unsigned char* chain_td_flags(const ChainTd* c);
const float* chain_td_history(ChainTd* c, int* Kp, hipStream_t st);
int  chain_fused_redo(ChainFused* c, const float* d_in, const float* d_hist, int hist_len, size_t n_samples, float* d_out, const unsigned char* d_flags, int flags_per_block, hipStream_t st);
int  chain_td_set_history256(ChainTd* c, const float* d_hist256, hipStream_t st);
void chain_td_destroy(ChainTd* c);
} // namespace gr4
int gr4hip_internal_fir_load_history(gr4hip_fir_t* f, const float* d_last256, hipStream_t st); // fir.hip
int gr4hip_internal_fir_set_guard_ratio(gr4hip_fir_t* f, double ratio);                          // fir.hip
// the kernel pair's filter marks its segments 6 dB earlier than a stand-alone fir_filter (15 dB below white noise's loss instead of 21): the split products' error, ~6e-6 of the
// OUTPUT at the FIR guard's own threshold, doubles on |Y|^2 (tools/fuzz_chain.py 120 41, case 828: a frame 33 dB down -- 19 dB more than noise loses -- at 1.015e-5)
#ifdef GR4_T_PAIR_GUARD_128 // developer build: the FIR guard's own threshold (what test_chain_kernel_pair_squares_its_filter_output pins: it fails with this)
constexpr double kChainPairGuardRatio = 1.0 / 128.0;
#else
constexpr double kChainPairGuardRatio = 1.0 / 32.0;
#endif
// (for the 33 .. 256-tap complex filter -- fir_mfma_f16x2_c32_kernel -- the tighter threshold is a SECOND verdict on the segment's whole output power; the quietest-column statistic keeps
// its 21 dB: on narrow-band noise -- 1 % pass band -- the quietest of sixteen columns alone sits below 15 dB in half of all segments and the float64 second evaluation ate a third of the
// settled stream's rate: 154 Gsamples/s with the two verdicts, 113 with one, 130 before the pair's filter was tightened at all)

using namespace gr4;

struct gr4hip_chain {
    size_t          ntaps = 0, N = 0;
    std::vector<float> taps;
    int             window = 0, algo = GR4HIP_CHAIN_UNFUSED;
    bool            auto_algo = false; // created with GR4HIP_CHAIN_AUTO (the guard belongs to AUTO only)
    gr4hip_fir_t*   fir = nullptr;
    gr4hip_fft_t*   fft = nullptr;
    gr4::ChainFused* fused = nullptr;
    gr4::ChainTd*   td = nullptr; // GR4HIP_CHAIN_FUSED_TD, or the guard's destination when the fft size allows
    gr4::ChainFused* td_redo = nullptr; // the tables of the same chain for chain_redo_kernel behind a judged chain_td launch (created on first use)
    DeviceBuffer    d_y;
    // dynamic-range guard (GR4HIP_CHAIN_AUTO on the fused kernel).  The fast-convolution kernels carry the float32 rounding of their transforms, ~2e-6 of the
    // INPUT rms per output sample (measured worst case ~1.3e-6); the parity bar is 1e-5 of the OUTPUT and the relative error of |Y|^2 is twice that of Y, so they meet it while
    // out_rms / in_rms >= 0.28, i.e. power ratio >= 0.08 (-11 dB; 0.04 until round 5: fuzz_chain found frames just above it at 1.02 - 1.29e-5).
    // The FIR-only fast-convolution paths of fir.hip (no squaring behind them) keep 0.04.
    // Every fused launch measures both powers of every frame (a frame below the threshold by itself marks the launch: chain_fused.hip).  The first call after create / reset probes its first frames synchronously; later calls
    // read the finished measurements of earlier ones without waiting.  Below the threshold the handle switches to the direct-form kernels (the
    // reference's own arithmetic) from the call that finds out onwards, until reset.
    bool            guard = false, probed = false, use_td = false;
    int             guard_mode = GR4HIP_GUARD_STRICT;
    float           last_ratio = -1.f; // most recent measured power ratio (< 0: none yet)
    float           last_marked = -1.f; // ... the fraction of that launch's frames the kernel marked (< 0: none yet) ...
    float           last_f64 = 0.f;     // ... and the fraction of all frames since create / reset that took the float64 evaluation: what a strict stream is moved on
    DeviceBuffer    d_hist_save;
    DeviceBuffer    d_multi;           // gr4hip_chain_process_multi on handles[0]: per-chain spectra when only their sum was asked for and one launch cannot fold them
};
// (round 6: the power ratio is reported, no longer decided on -- a frame is judged on the fourth-moment statistic of chain_fused.hip, kGuardR4Max)
// When does a stream move to the time-domain kernel pair (float32 products: 69 Gsamples/s) for good?  Strict guard: the marked frames are evaluated again behind their launch anyway;
// a frame that chain_td16_kernel settles costs 1 / 333 + 1 / 113 ns per sample (85 Gsamples/s if every frame is marked: faster than the pair), a frame that goes on to float64
// 1 / 34 more -- the pair is the faster way from one float64 frame in ten on.  Deferred guard: nothing is evaluated again -- one marked frame moves the stream, as before.
constexpr float  kGuardMoveFloat64Fraction = 0.1f;
static bool chain_should_move(const gr4hip_chain* c) { return c->guard_mode == GR4HIP_GUARD_STRICT ? c->last_f64 > kGuardMoveFloat64Fraction : c->last_marked > 0.f; }
static void chain_note_measurement(gr4hip_chain* c, gr4::ChainFused* f, bool wait) {
    float r, m, f64;
    if (chain_fused_power_ratio(f, wait, false, &r, &m, &f64)) { c->last_ratio = r; c->last_marked = m; c->last_f64 = f64; }
}
constexpr size_t kGuardProbeFrames   = 8; // in units of 8192-sample blocks
constexpr size_t kTdAutoMaxTaps      = 64; // AUTO: up to here the fused time-domain kernel beats the fused fast convolution (tools/chain_modes_rates.py)

extern "C" {

int gr4hip_chain_create(gr4hip_chain_t** out, const float* h_taps, size_t ntaps, size_t fft_size, int window, int algo) {
    GR4_REQUIRE(out, "chain: null output handle");
    GR4_REQUIRE(h_taps && ntaps >= 1, "chain: need at least one tap");
    GR4_REQUIRE(algo >= GR4HIP_CHAIN_AUTO && algo <= GR4HIP_CHAIN_TIME_DOMAIN, "chain: unknown algo %d", algo);
    auto* c = new (std::nothrow) gr4hip_chain();
    GR4_REQUIRE(c, "out of host memory");
    c->ntaps  = ntaps;
    c->taps.assign(h_taps, h_taps + ntaps);
    c->N      = fft_size;
    c->window = window;
    int use   = algo;
    if (algo == GR4HIP_CHAIN_AUTO) {
        use = GR4HIP_CHAIN_UNFUSED;
        if (ntaps <= kTdAutoMaxTaps && chain_td_supported(ntaps, fft_size, window)) use = GR4HIP_CHAIN_FUSED_TD;
        else if (chain_fused_supported(ntaps, fft_size, window, GR4HIP_CHAIN_FUSED_FD)) use = GR4HIP_CHAIN_FUSED_FD;
    } else if (algo == GR4HIP_CHAIN_FUSED_TD ? !chain_td_supported(ntaps, fft_size, window)
                                             : (algo != GR4HIP_CHAIN_UNFUSED && algo != GR4HIP_CHAIN_TIME_DOMAIN && !chain_fused_supported(ntaps, fft_size, window, algo))) {
        set_error("chain: fused algo %d does not support ntaps=%zu fft_size=%zu window=%d", algo, ntaps, fft_size, window);
        delete c;
        return GR4HIP_UNSUPPORTED;
    }
    c->algo  = use;
    c->auto_algo = algo == GR4HIP_CHAIN_AUTO;
    c->guard = algo == GR4HIP_CHAIN_AUTO && use == GR4HIP_CHAIN_FUSED_FD && ntaps > 1;
    int rc;
    if (use == GR4HIP_CHAIN_UNFUSED || use == GR4HIP_CHAIN_TIME_DOMAIN) {
        rc = gr4hip_fir_create(&c->fir, GR4HIP_C32, h_taps, ntaps, 1);
        // float32 PRODUCTS for both (round 6's last day; until then GR4HIP_CHAIN_UNFUSED -- what GR4HIP_CHAIN_AUTO takes at fft sizes beyond 8192, other than powers of two, or
        // past 256 taps -- ran the two-term f16 direct form under its two power verdicts).  A power statistic cannot see what the transform behind the filter does with the 22-bit
        // products' error: it is COHERENT on a tone (its residue behind the filter is off by up to 2^-22 sum|b| / |H(f)| of itself), and the transform gathers it into the one bin
        // where the metric looks.  tools/dbg/pair_coherent.py -- 65 taps, a tone 15 dB above the noise removed by ~60 dB, 16384 points: 28 of 2 400 streams at 1.0 .. 4.2e-5, the
        // reference's float32 sum at 6e-7.  fir_filter by itself keeps the f16 kernels: its metric is the time-domain sample against the output's rms, where that error is 1e-6.
        // (Measured instead: the f16 kernels with EVERY segment evaluated again in float64 behind them -- 27 Gsamples/s at 256 taps x 16384 points against 44 on float32 products.)
        if (!rc) rc = gr4hip_fir_set_algo(c->fir, GR4HIP_FIR_TIME_DOMAIN_F32);
        if (!rc) rc = gr4hip_internal_fir_set_guard_ratio(c->fir, kChainPairGuardRatio);
        if (!rc) rc = gr4hip_fft_create(&c->fft, GR4HIP_C32, fft_size, window, 0);
    } else if (use == GR4HIP_CHAIN_FUSED_TD) {
        rc = chain_td_create(&c->td, h_taps, ntaps, fft_size, window);
    } else {
        rc = chain_fused_create(&c->fused, h_taps, ntaps, fft_size, window, use);
        if (!rc && c->guard) chain_fused_set_measure(c->fused, true);
    }
    if (rc) { gr4hip_chain_destroy(c); return rc; }
    *out = c;
    return GR4HIP_OK;
}

int gr4hip_chain_reset(gr4hip_chain_t* c) {
    GR4_REQUIRE(c, "chain_reset: null handle");
    c->probed = c->use_td = false;
    c->last_ratio = c->last_marked = -1.f;
    c->last_f64 = 0.f;
    if (c->fused && c->fir) { int rc = gr4hip_fir_reset(c->fir); if (rc) return rc; }
    if (c->td) { int rc = chain_td_reset(c->td); if (rc || !c->fused) return rc; }
    return c->fused ? chain_fused_reset(c->fused) : gr4hip_fir_reset(c->fir);
}

// the fused time-domain kernel under the guard every FIR kernel answers to (round 5): its float32 sums -- exact products, the matrix pipe's order -- measured up to 4.6 x the
// reference-order float32 sum's error under a rejected interferer (tools/fuzz_chain.py against the oracle).  It marks the 4096-sample segments whose filter output carries
// less than (sum b^2 / 128) x their input power; chain_redo_kernel evaluates the blocks that hold one again (float64 products) behind the launch.  No host in the loop.
static int chain_td_run(gr4hip_chain* c, const void* d_in, size_t frames, float* d_mag2, gr4hip_stream_t stream) {
    hipStream_t st     = as_stream(stream);
    bool        judged = c->guard_mode != GR4HIP_GUARD_OFF && c->taps.size() > 1 && chain_fused_supported(c->taps.size(), c->N, c->window, GR4HIP_CHAIN_FUSED_FD);
    if (judged && !c->td_redo) {
        if (const int rc = chain_fused_create(&c->td_redo, c->taps.data(), c->taps.size(), c->N, c->window, GR4HIP_CHAIN_FUSED_FD)) return rc;
    }
    int          Kp   = 0;
    const float* hist = chain_td_history(c->td, &Kp, st); // the samples in front of THIS call (the launch writes the other half of the pair)
    if (!hist) return GR4HIP_RUNTIME_ERROR;
    int rc = chain_td_process(c->td, static_cast<const float*>(d_in), frames, d_mag2, st, judged);
    if (!rc && judged) rc = chain_fused_redo(c->td_redo, static_cast<const float*>(d_in), hist, Kp, frames * c->N, d_mag2, chain_td_flags(c->td), 2, st);
    return rc;
}

// direct-form FIR kernel -> y in HBM -> FFT kernel (the GR4HIP_CHAIN_TIME_DOMAIN path; also where the guard sends a fused AUTO chain)
// (Measured and dropped: four batches on two internal streams so that the FIR of batch b + 1 -- matrix pipe -- runs beside the FFT of batch b: 93 instead of 98
// Gsamples/s at 256 taps, 167 instead of 179 at 64: two grids that each fill the chip take turns anyway, and the extra launches cost.)
static int chain_time_domain(gr4hip_chain* c, const void* d_in, size_t frames, float* d_mag2, gr4hip_stream_t stream) {
    if (c->td) return chain_td_run(c, d_in, frames, d_mag2, stream); // one launch where the size allows
    const size_t n  = frames * c->N;
    int          rc = c->d_y.ensure(n * 2 * sizeof(float));
    if (rc) return rc;
    rc = gr4hip_fir_process(c->fir, d_in, n, c->d_y.ptr, nullptr, stream);
    if (rc) return rc;
    return gr4hip_fft_mag2(c->fft, c->d_y.ptr, frames, d_mag2, stream);
}

// the guard found the filter removing most of the input: from here on the stream runs through the direct-form kernels, starting from `d_hist256`
static int chain_switch_to_time_domain(gr4hip_chain* c, const float* d_hist256, hipStream_t st) {
    std::vector<float>& taps = c->taps;
    int rc = GR4HIP_OK;
    if (!d_hist256) return GR4HIP_RUNTIME_ERROR; // (chain_fused_history could not enqueue the pending reset: the error text is set)
    // the kernel pair with float32 products (GR4HIP_FIR_TIME_DOMAIN_F32), at every fft size: the regime that trips the guard -- a rejected signal far above the
    // output -- is the one in which the three-term bf16 products of chain_td_kernel / the split-product direct forms measure 3 .. 16 x a float32 sum's error.
    // (Round 4 tried the two-term f16 direct form here, whose own guard redoes the segments that reject more than 36 dB of their power: the pair went from 97 to
    // ~160 Gsamples/s, but between the two thresholds -- 14 .. 36 dB rejected -- its error, small against the filtered samples, is a function of the rejected tone and
    // the transform behind it gathers it into a few bins: 4 of the chain guard tests failed the bar there.  The float32 products stay.)
    if (!c->fir) {
        rc = gr4hip_fir_create(&c->fir, GR4HIP_C32, taps.data(), taps.size(), 1);
        // float32 PRODUCTS (the f32 matrix pipe: the reference's own arithmetic) -- round 4's choice, again since round 6.  Round 5 took the two-term f16 direct form with its
        // per-segment guard here (140 instead of 69 Gsamples/s); tools/fuzz_chain.py "wide" (round 6: interferers from -5 dB, anywhere outside the pass band) found what a power
        // statistic cannot see: the 22-bit products' error is COHERENT on a tone -- its residue behind the filter is off by 2^-22 sum|b| / |H(f)| of itself --, and the transform
        // gathers it into the one bin where the metric looks: 1.2e-5 at a residue that reaches the rms level of the output spectrum 26 dB or more down.
        if (!rc) rc = gr4hip_fir_set_algo(c->fir, GR4HIP_FIR_TIME_DOMAIN_F32);
        if (!rc) rc = gr4hip_internal_fir_set_guard_ratio(c->fir, kChainPairGuardRatio);
        if (!rc) rc = gr4hip_fft_create(&c->fft, GR4HIP_C32, c->N, c->window, 0);
    }
    if (!rc) rc = gr4hip_internal_fir_load_history(c->fir, d_hist256, st);
    if (!rc) c->use_td = true;
    return rc;
}

int gr4hip_chain_process(gr4hip_chain_t* c, const void* d_in, size_t n_samples, float* d_mag2, size_t* n_frames_p, gr4hip_stream_t stream) {
    GR4_REQUIRE(c, "chain_process: null handle");
    const size_t frames = n_samples / c->N; // the FFT block consumes whole frames only (fft.hpp:131-134)
    if (n_frames_p) *n_frames_p = frames;
    if (frames == 0) return n_samples ? GR4HIP_INSUFFICIENT_INPUT : GR4HIP_OK;
    GR4_REQUIRE(d_in && d_mag2, "chain_process: null device pointer");
    if (c->td && !c->fused) return chain_td_run(c, d_in, frames, d_mag2, stream);
    if (c->fused && !c->guard) return chain_fused_process(c->fused, static_cast<const float*>(d_in), frames, d_mag2, as_stream(stream));
    if (c->fused) { // GR4HIP_CHAIN_AUTO on the fused kernel: dynamic-range guard
        hipStream_t  st  = as_stream(stream);
        const float* x   = static_cast<const float*>(d_in);
        const size_t per = c->N < 8192 ? 8192 / c->N : 1; // fft frames per 8192-sample block
        if (c->use_td) return chain_time_domain(c, d_in, frames, d_mag2, stream);
        if (c->guard_mode == GR4HIP_GUARD_STRICT) {
            // nothing out of tolerance is ever published, and nobody waits: the fused kernel marks the frames whose output power fell below the threshold, and
            // chain_redo_kernel -- enqueued behind it on the same stream -- evaluates exactly those frames again in the time domain (float64 products) over the
            // fused results.  The call returns when both launches are enqueued ("user code must not block in work()", docs/USER_API_advanced_work.md).  Until
            // round 5 the call spun on a mapped word until its launch had ended and redid the whole span on the time-domain kernel pair.
            // The measurement of an EARLIER launch, when it has arrived, still moves a stream that rejects most of its input to the time-domain kernels for good
            // (they are the faster way through such a stream than fused kernel + second evaluation of every frame) -- without waiting for anything.
            chain_note_measurement(c, c->fused, false);
            if (chain_should_move(c)) {
                int rc = chain_switch_to_time_domain(c, chain_fused_history(c->fused, st), st);
                if (rc) return rc;
                return chain_time_domain(c, d_in, frames, d_mag2, stream);
            }
            chain_fused_set_redo(c->fused, true);
            return chain_fused_process(c->fused, x, frames, d_mag2, st);
        }
        chain_fused_set_redo(c->fused, false);
        chain_note_measurement(c, c->fused, false); // an earlier launch has finished: no waiting
        size_t done = 0;
        if (!c->probed) { // first call after create / reset: the first blocks synchronously, before the rest of the span is committed to an algorithm
            int rc = c->d_hist_save.ensure(256 * 2 * sizeof(float));
            if (rc) return rc;
            GR4_HIP_TRY(hipMemcpyAsync(c->d_hist_save.ptr, chain_fused_history(c->fused, st), 256 * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
            const size_t probe = std::min(frames, kGuardProbeFrames * per);
            rc = chain_fused_process(c->fused, x, probe, d_mag2, st);
            if (rc) return rc;
            chain_note_measurement(c, c->fused, true);
            c->probed = true;
            done      = probe;
            if (chain_should_move(c)) { // redo the probed frames too, from the history the call started with
                rc = chain_switch_to_time_domain(c, static_cast<const float*>(c->d_hist_save.ptr), st);
                if (rc) return rc;
                return chain_time_domain(c, d_in, frames, d_mag2, stream);
            }
        } else if (chain_should_move(c)) { // an earlier call ran into the regime: switch before this one
            int rc = chain_switch_to_time_domain(c, chain_fused_history(c->fused, st), st);
            if (rc) return rc;
            return chain_time_domain(c, d_in, frames, d_mag2, stream);
        }
        if (done == frames) return GR4HIP_OK;
        return chain_fused_process(c->fused, x + done * c->N * 2, frames - done, d_mag2 + done * c->N, st);
    }
    return chain_time_domain(c, d_in, frames, d_mag2, stream);
}

int gr4hip_chain_set_guard_mode(gr4hip_chain_t* c, int mode) {
    GR4_REQUIRE(c, "chain_set_guard_mode: null handle");
    GR4_REQUIRE(mode >= GR4HIP_GUARD_STRICT && mode <= GR4HIP_GUARD_OFF, "chain_set_guard_mode: unknown mode %d", mode);
    c->guard_mode = mode;
    if (c->fused) {
        const bool was = c->guard;
        c->guard = mode != GR4HIP_GUARD_OFF && c->algo == GR4HIP_CHAIN_FUSED_FD && c->auto_algo && c->ntaps > 1;
        if (was != c->guard) chain_fused_set_measure(c->fused, c->guard);
    }
    return GR4HIP_OK;
}

// Several chains, one launch (include/gr4hip.h).  The single launch applies when every chain runs the fused frequency-domain kernel at 8192 points with the
// rectangular window and none has been moved to the time domain by its guard; anything else is served chain by chain (+ math::Add) with the same results.
int gr4hip_chain_process_multi(gr4hip_chain_t* const* chains, size_t n_chains, const void* const* d_in, size_t n_samples, float* const* d_mag2, float* d_sum,
                               size_t* n_frames_p, gr4hip_stream_t stream) {
    GR4_REQUIRE(chains && n_chains >= 1 && n_chains <= 16, "chain_process_multi: 1 .. 16 chains");
    GR4_REQUIRE(d_in && (d_mag2 || d_sum), "chain_process_multi: null argument");
    for (size_t i = 0; i < n_chains; ++i) GR4_REQUIRE(chains[i] && chains[i]->N == chains[0]->N, "chain_process_multi: null handle or differing fft sizes");
    gr4hip_chain* c0     = chains[0];
    const size_t  frames = n_samples / c0->N;
    if (n_frames_p) *n_frames_p = frames;
    if (frames == 0) return n_samples ? GR4HIP_INSUFFICIENT_INPUT : GR4HIP_OK;
    for (size_t i = 0; i < n_chains; ++i) GR4_REQUIRE(d_in[i] && (!d_mag2 || d_mag2[i]), "chain_process_multi: null device pointer");
    hipStream_t st = as_stream(stream);
    bool one_launch = c0->N == 8192, shared = true, any_guard = false;
    for (size_t i = 0; i < n_chains; ++i) {
        gr4hip_chain* c = chains[i];
        one_launch = one_launch && c->fused && !c->use_td && chain_fused_multi_capable(c->fused);
        shared     = shared && c->taps == c0->taps;
        any_guard  = any_guard || c->guard;
    }
    // the combiner in registers; otherwise per-chain spectra (+ the n-ary Add below).  The fold measures all channels together into chain 0's slots: with a guard on
    // some chain but not on chain 0 nothing would be measured at all, so such a mix keeps its per-chain spectra (every guarded chain then measures itself)
    const bool fold = d_sum && !d_mag2 && shared && n_chains > 1 && (!any_guard || c0->guard);
    // per-chain spectra nobody asked for but the fold needs: scratch on handles[0]
    std::vector<float*> outs(n_chains, nullptr);
    if (!fold) {
        if (d_mag2) for (size_t i = 0; i < n_chains; ++i) outs[i] = d_mag2[i];
        else {
            int rc = c0->d_multi.ensure(n_chains * frames * c0->N * sizeof(float));
            if (rc) return rc;
            for (size_t i = 0; i < n_chains; ++i) outs[i] = static_cast<float*>(c0->d_multi.ptr) + i * frames * c0->N;
        }
    }
    auto chain_by_chain = [&]() -> int {
        std::vector<float*> o = outs;
        if (fold) { // the fold needs the spectra after all
            int rc = c0->d_multi.ensure(n_chains * frames * c0->N * sizeof(float));
            if (rc) return rc;
            for (size_t i = 0; i < n_chains; ++i) o[i] = static_cast<float*>(c0->d_multi.ptr) + i * frames * c0->N;
        }
        for (size_t i = 0; i < n_chains; ++i) {
            int rc = gr4hip_chain_process(chains[i], d_in[i], n_samples, o[i], nullptr, stream);
            if (rc) return rc;
        }
        if (!d_sum) return GR4HIP_OK;
        std::vector<const void*> ins(o.begin(), o.end());
        return gr4hip_math_nary(GR4HIP_ADD, GR4HIP_F32, ins.data(), n_chains, d_sum, frames * c0->N, stream);
    };
    if (!one_launch) return chain_by_chain();

    std::vector<gr4::ChainFused*> fs(n_chains);
    std::vector<const float*>     xs(n_chains);
    for (size_t i = 0; i < n_chains; ++i) { fs[i] = chains[i]->fused; xs[i] = static_cast<const float*>(d_in[i]); }
    bool strict = false; // the most conservative mode among the guarded chains decides for the launch they share
    for (size_t i = 0; i < n_chains; ++i) strict = strict || (chains[i]->guard && chains[i]->guard_mode == GR4HIP_GUARD_STRICT);
    // a span the guard rejects is redone from the histories the call started with: guarded chains move to the time domain, the others (GUARD_OFF, explicit
    // FUSED_FD: "never switches") stay on the fused kernel -- whose history the rejected launch has already advanced, so it is put back first
    const auto prepare_redo = [&]() -> int {
        for (size_t i = 0; i < n_chains; ++i) {
            gr4hip_chain* c = chains[i];
            if (c->guard) { if (const int rc = chain_switch_to_time_domain(c, static_cast<const float*>(c->d_hist_save.ptr), st)) return rc; }
            else if (const int rc = chain_fused_set_history(c->fused, static_cast<const float*>(c->d_hist_save.ptr), st)) return rc;
        }
        return GR4HIP_OK;
    };
    if (any_guard) { // the histories the call starts with (a span the guard rejects is redone from them)
        for (size_t i = 0; i < n_chains; ++i) {
            gr4hip_chain* c = chains[i];
            int rc = c->d_hist_save.ensure(256 * 2 * sizeof(float));
            if (rc) return rc;
            GR4_HIP_TRY(hipMemcpyAsync(c->d_hist_save.ptr, chain_fused_history(c->fused, st), 256 * 2 * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
        { // a finished earlier launch decides for this one, without waiting (strict: the frames THIS launch marks are evaluated again on the device behind it)
            bool bad = false;
            for (size_t i = 0; i < n_chains; ++i) {
                gr4hip_chain* c = chains[fold ? 0 : i];
                chain_note_measurement(c, c->fused, false);
                bad = bad || (chains[i]->guard && chain_should_move(c));
            }
            if (bad) { // (nothing launched yet in this call: the saved histories are the current ones, the copy back is a no-op)
                if (const int rc = prepare_redo()) return rc;
                return chain_by_chain();
            }
        }
    }
    // (round 5) strict: nobody waits -- the launch writes a verdict byte per frame (the fold: a frame is marked when any channel's share of it fell below the threshold) and
    // the second evaluation rides the same stream (chain_redo_kernel per chain, chain_redo_fold_kernel for the fold); until round 5 the call awaited its launch's
    // measurement and redid the whole span chain by chain on the host's say-so
    int rc = chain_fused_process_multi(fs.data(), n_chains, shared, xs.data(), frames, fold ? nullptr : outs.data(), fold ? d_sum : nullptr, st, strict);
    if (rc) return rc;
    for (size_t i = 0; i < n_chains; ++i) chains[i]->probed = true;
    if (!fold && d_sum) {
        std::vector<const void*> ins(outs.begin(), outs.end());
        return gr4hip_math_nary(GR4HIP_ADD, GR4HIP_F32, ins.data(), n_chains, d_sum, frames * c0->N, stream);
    }
    return GR4HIP_OK;
}

int gr4hip_chain_last_power_ratio(gr4hip_chain_t* c, float* ratio, int* time_domain, gr4hip_stream_t stream) {
    GR4_REQUIRE(c && ratio, "chain_last_power_ratio: null argument");
    (void)stream;
    if (c->fused && c->guard) chain_note_measurement(c, c->fused, true); // waits for the last measured launch
    *ratio = c->last_ratio;
    if (time_domain) *time_domain = c->use_td ? 1 : 0;
    return GR4HIP_OK;
}

int gr4hip_chain_last_guard_fractions(gr4hip_chain_t* c, float* marked, float* float64, gr4hip_stream_t stream) {
    GR4_REQUIRE(c && marked && float64, "chain_last_guard_fractions: null argument");
    (void)stream;
    if (c->fused && c->guard) chain_note_measurement(c, c->fused, true); // waits for the last measured launch
    *marked  = c->last_marked;
    *float64 = c->last_f64;
    return GR4HIP_OK;
}

int gr4hip_chain_set_max_workgroups(gr4hip_chain_t* c, unsigned n) {
    GR4_REQUIRE(c, "chain_set_max_workgroups: null handle");
    if (c->fused) chain_fused_set_max_workgroups(c->fused, n);
    return GR4HIP_OK;
}

int gr4hip_chain_get_algo(const gr4hip_chain_t* c, int* algo) { GR4_REQUIRE(c && algo, "chain_algo: null"); *algo = c->algo; return GR4HIP_OK; }

int gr4hip_chain_destroy(gr4hip_chain_t* c) {
    if (!c) return GR4HIP_OK;
    if (c->fir) gr4hip_fir_destroy(c->fir);
    if (c->fft) gr4hip_fft_destroy(c->fft);
    if (c->fused) chain_fused_destroy(c->fused);
    if (c->td) chain_td_destroy(c->td);
    if (c->td_redo) chain_fused_destroy(c->td_redo);
    delete c;
    return GR4HIP_OK;
}

} // extern "C"
