// fft_fast_pk.hip -- the fft_fast_kernel instantiations that are FASTER with hipcc's SLP vectoriser (packed f32 math): N = 256 and 8192.
// build.sh compiles this file without -fno-slp-vectorize; see fft_kernels.hpp.
// (round 5) streaming (nt) hints on the frames' loads and the results' stores, like fft.hip (profiles/r05_streaming_hints.txt: N = 256 |X|^2 442 - 456 -> 472 - 481 Gsamples/s)
#ifndef GR4_BUF_STORE_AUX
#define GR4_BUF_STORE_AUX 2
#endif
#ifndef GR4_BUF_LOAD_AUX
#define GR4_BUF_LOAD_AUX 2
#endif
#include "fft_kernels.hpp"

namespace gr4 {
int fft_fast_launch_256(const float* d_in, const float* d_window, const float2* d_tw, const FftOutputs& o, long n_frames, hipStream_t st) {
    return fft_fast_launch<8>(d_in, d_window, d_tw, o, n_frames, st);
}
int fft_fast_launch_8192(const float* d_in, const float* d_window, const float2* d_tw, const FftOutputs& o, long n_frames, hipStream_t st) {
    return fft_fast_launch<13>(d_in, d_window, d_tw, o, n_frames, st);
}
} // namespace gr4
