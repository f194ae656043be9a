// chain.hip -- the headline chain: complex<float> fir_filter -> FFT block frames -> |X|^2 (BASELINE.json configs[1]).
//
// GR4HIP_CHAIN_UNFUSED: gr4hip_fir_process -> y in HBM -> gr4hip_fft_mag2 (any size the FFT block supports).
// GR4HIP_CHAIN_FUSED_FD (chain_fused.hip): one persistent launch, any window, fft_size 256...8192, <= 256 taps; AUTO picks it when it applies.
#include "common.hpp"

namespace gr4 {
int chain_fused_supported(size_t ntaps, size_t fft_size, int window, int algo);
struct ChainFused;
int  chain_fused_create(ChainFused** out, const float* taps, size_t ntaps, size_t fft_size, int window, int algo);
int  chain_fused_reset(ChainFused* c);
int  chain_fused_process(ChainFused* c, const float* d_in, size_t n_frames, float* d_mag2, hipStream_t st);
void chain_fused_destroy(ChainFused* c);
void chain_fused_set_max_workgroups(ChainFused* c, unsigned n);
} // namespace gr4

using namespace gr4;

struct gr4hip_chain {
    size_t          ntaps = 0, N = 0;
    int             window = 0, algo = GR4HIP_CHAIN_UNFUSED;
    gr4hip_fir_t*   fir = nullptr;
    gr4hip_fft_t*   fft = nullptr;
    gr4::ChainFused* fused = nullptr;
    DeviceBuffer    d_y;
};

extern "C" {

int gr4hip_chain_create(gr4hip_chain_t** out, const float* h_taps, size_t ntaps, size_t fft_size, int window, int algo) {
    GR4_REQUIRE(out, "chain: null output handle");
    GR4_REQUIRE(h_taps && ntaps >= 1, "chain: need at least one tap");
    GR4_REQUIRE(algo >= GR4HIP_CHAIN_AUTO && algo <= GR4HIP_CHAIN_TIME_DOMAIN, "chain: unknown algo %d", algo);
    auto* c = new (std::nothrow) gr4hip_chain();
    GR4_REQUIRE(c, "out of host memory");
    c->ntaps  = ntaps;
    c->N      = fft_size;
    c->window = window;
    int use   = algo;
    if (algo == GR4HIP_CHAIN_AUTO) {
        use = GR4HIP_CHAIN_UNFUSED;
        for (int cand : {GR4HIP_CHAIN_FUSED_FD, GR4HIP_CHAIN_FUSED_TD})
            if (chain_fused_supported(ntaps, fft_size, window, cand)) { use = cand; break; }
    } else if (algo != GR4HIP_CHAIN_UNFUSED && algo != GR4HIP_CHAIN_TIME_DOMAIN && !chain_fused_supported(ntaps, fft_size, window, algo)) {
        set_error("chain: fused algo %d does not support ntaps=%zu fft_size=%zu window=%d", algo, ntaps, fft_size, window);
        delete c;
        return GR4HIP_UNSUPPORTED;
    }
    c->algo = use;
    int rc;
    if (use == GR4HIP_CHAIN_UNFUSED || use == GR4HIP_CHAIN_TIME_DOMAIN) {
        rc = gr4hip_fir_create(&c->fir, GR4HIP_C32, h_taps, ntaps, 1);
        if (!rc && use == GR4HIP_CHAIN_TIME_DOMAIN) rc = gr4hip_fir_set_algo(c->fir, GR4HIP_FIR_TIME_DOMAIN);
        if (!rc) rc = gr4hip_fft_create(&c->fft, GR4HIP_C32, fft_size, window, 0);
    } else {
        rc = chain_fused_create(&c->fused, h_taps, ntaps, fft_size, window, use);
    }
    if (rc) { gr4hip_chain_destroy(c); return rc; }
    *out = c;
    return GR4HIP_OK;
}

int gr4hip_chain_reset(gr4hip_chain_t* c) {
    GR4_REQUIRE(c, "chain_reset: null handle");
    return c->fused ? chain_fused_reset(c->fused) : gr4hip_fir_reset(c->fir);
}

int gr4hip_chain_process(gr4hip_chain_t* c, const void* d_in, size_t n_samples, float* d_mag2, size_t* n_frames_p, gr4hip_stream_t stream) {
    GR4_REQUIRE(c, "chain_process: null handle");
    const size_t frames = n_samples / c->N; // the FFT block consumes whole frames only (fft.hpp:131-134)
    if (n_frames_p) *n_frames_p = frames;
    if (frames == 0) return n_samples ? GR4HIP_INSUFFICIENT_INPUT : GR4HIP_OK;
    GR4_REQUIRE(d_in && d_mag2, "chain_process: null device pointer");
    const size_t n = frames * c->N;
    if (c->fused) return chain_fused_process(c->fused, static_cast<const float*>(d_in), frames, d_mag2, as_stream(stream));
    int rc = c->d_y.ensure(n * 2 * sizeof(float));
    if (rc) return rc;
    rc = gr4hip_fir_process(c->fir, d_in, n, c->d_y.ptr, nullptr, stream);
    if (rc) return rc;
    return gr4hip_fft_mag2(c->fft, c->d_y.ptr, frames, d_mag2, stream);
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
    delete c;
    return GR4HIP_OK;
}

} // extern "C"
