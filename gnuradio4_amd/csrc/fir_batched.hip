// fir_batched.hip -- many-channel FIR as a block-Toeplitz contraction on the f32 MFMA units (configs[3]).
#include "common.hpp"
using namespace gr4;
struct gr4hip_fir_batched { int dummy; };
extern "C" {
int gr4hip_fir_batched_create(gr4hip_fir_batched_t**, size_t, const float*, size_t) { set_error("fir_batched: not implemented yet"); return GR4HIP_UNSUPPORTED; }
int gr4hip_fir_batched_reset(gr4hip_fir_batched_t*) { return GR4HIP_UNSUPPORTED; }
int gr4hip_fir_batched_process(gr4hip_fir_batched_t*, const float*, size_t, size_t, float*, size_t, gr4hip_stream_t) { return GR4HIP_UNSUPPORTED; }
int gr4hip_fir_batched_destroy(gr4hip_fir_batched_t*) { return GR4HIP_OK; }
}
