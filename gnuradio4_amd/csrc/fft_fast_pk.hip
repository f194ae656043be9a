// fft_fast_pk.hip -- the fft_fast_kernel instantiations that are FASTER with hipcc's SLP vectoriser (packed f32 math): N = 256 and 8192.
// build.sh compiles this file without -fno-slp-vectorize; see fft_kernels.hpp.
#include "fft_kernels.hpp"

namespace gr4 {
int fft_fast_launch_256(const float* d_in, const float* d_window, const float2* d_tw, const FftOutputs& o, long n_frames, hipStream_t st) {
    return fft_fast_launch<8>(d_in, d_window, d_tw, o, n_frames, st);
}
int fft_fast_launch_8192(const float* d_in, const float* d_window, const float2* d_tw, const FftOutputs& o, long n_frames, hipStream_t st) {
    return fft_fast_launch<13>(d_in, d_window, d_tw, o, n_frames, st);
}
} // namespace gr4
