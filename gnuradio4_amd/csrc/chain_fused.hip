// chain_fused.hip -- fused FIR->FFT->mag2 kernels (placeholder: selection hooks; kernels land in the next milestone).
#include "common.hpp"

namespace gr4 {
struct ChainFused {};
int  chain_fused_supported(size_t, size_t, int, int) { return 0; }
int  chain_fused_create(ChainFused**, const float*, size_t, size_t, int, int) { set_error("fused chain not available"); return GR4HIP_UNSUPPORTED; }
int  chain_fused_reset(ChainFused*) { return GR4HIP_OK; }
int  chain_fused_process(ChainFused*, const float*, size_t, float*, hipStream_t) { return GR4HIP_UNSUPPORTED; }
void chain_fused_destroy(ChainFused*) {}
} // namespace gr4
