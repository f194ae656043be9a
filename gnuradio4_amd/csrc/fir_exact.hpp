// fir_exact.hpp -- the second evaluation behind every FIR / decimator kernel (fir_exact.hip): declarations shared by fir.hip, fir_batched.hip
#pragma once
#include "common.hpp"
#include "ewise.hpp"

namespace gr4 {

// evaluates the marked outputs (flags: one byte per 2^seg_shift outputs; null: every output) of  y[o] = sum_k b[k] x[D o - k],  o < n_out,  on the FP64 matrix pipe.
// Sizes in samples (float, or complex when cplx); strides in floats.  gate (optional): the launch does nothing unless *gate != 0.  pre / post (optional): the main kernel's
// load / store programs (positions as the main kernel takes them: pre.pos = stream index of x[0], post.pos = stream index of y[0]).
// GR4HIP_UNSUPPORTED: the staged unit (decimation x taps) does not fit the CU's LDS.
int fir_exact_launch(const float* x, long n_in, const float* hist, int Kh, const float* d_taps, int ntaps, int D, int cplx, float* y, long n_out, const unsigned char* flags, int seg_shift,
                     const unsigned* gate, hipStream_t st, unsigned nch = 1, long in_stride = 0, long out_stride = 0, long taps_stride = 0, long flags_stride = 0, const EwiseHook* pre = nullptr,
                     const EwiseHook* post = nullptr);
// the verdict for a kernel that does not judge itself: flags[s] != 0 where the 2^seg_shift outputs of segment s carry less than gthr x the power of the samples in front of them
// gthr_all > 0: a second verdict on the segment's WHOLE output power (the chain's kernel pair: chain.hip kChainPairGuardRatio)
int fir_judge_launch(const float* x, long n_in, const float* y, long n_out, int D, int cplx, int seg_shift, float gthr, unsigned char* flags, hipStream_t st, float gthr_all = 0.f);

} // namespace gr4
