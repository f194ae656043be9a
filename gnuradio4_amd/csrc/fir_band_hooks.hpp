// fir_band_hooks.hpp -- the load / store programs of the band-form decimators (fir_bf16.hip, fir_decim_f16.hip): shared device helpers
#pragma once
#include "common.hpp"
#include "ewise.hpp"

namespace gr4 {

// ---- the neighbours of a decimator in its launch (gr4hip_fir_set_prologue / _epilogue; fir.hip): a program applied to the samples on their way into the bf16 planes
// (positions are FLOAT indices of the stream as the kernel sees it; cplx: two floats per sample, programs of complex<float> items) and to the outputs before the store
struct BdHooks {
    EwiseHook pre, post;
    int       cplx = 0;
    // the guard (fir_f16.hip's, per segment): the three-term products' error is relative to the PRODUCTS, so a segment whose output power is below gthr x its staged input
    // power -- more than 21 dB rejected beyond what white noise would lose -- is MARKED in flags[segment], and fir_exact_kernel (fir_exact.hip), launched behind this
    // kernel, evaluates it again on the FP64 matrix pipe (hooked launches: the powers are those of the prologue's output and of the filter's, and the second evaluation
    // runs the same programs).  flags == nullptr: nobody judges
    float          gthr  = 0.f;
    unsigned char* flags = nullptr;
};
// the judge's two sums of a segment: every wave leaves its part in st[0 .. 3] (input) / st[4 .. 7] (output); after the segment's last barrier thread 0 decides.  The
// quietest wave's outputs count as the segment's (a start-up transient in one quarter of it does not hide that the rest is all rejection)
__device__ __forceinline__ void bd_judge(const float* st, const BdHooks& hk, long sg, int D) {
    const float px = (st[0] + st[1]) + (st[2] + st[3]);
    const float py = 4.f * __builtin_fminf(__builtin_fminf(st[4], st[5]), __builtin_fminf(st[6], st[7]));
    hk.flags[sg]   = (py * (float)D < hk.gthr * px) ? 3 : 0; // (a NaN power compares false: unmarked -- the non-finite classes are these kernels' own)
}
__device__ __forceinline__ float4 bd_hook4(float4 v, const EwiseHook& h, int cplx, long fi /*float index of v.x: a multiple of 4*/) {
    if (cplx) {
        float2 e[2] = {make_float2(v.x, v.y), make_float2(v.z, v.w)};
        ewise_hook<float2, 2>(e, h, fi >> 1);
        return make_float4(e[0].x, e[0].y, e[1].x, e[1].y);
    }
    float e[4] = {v.x, v.y, v.z, v.w};
    ewise_hook<float, 4>(e, h, fi);
    return make_float4(e[0], e[1], e[2], e[3]);
}
// The load program is ONE rotator on a complex stream and nothing rides on the store (EwiseHook::rotor_only: rotator -> decimating FIR, a down-converter): a lane's
// float4 number u of a segment holds the complex samples (in0 + 4 (tid + 256 u)) / 2 and the one behind it -- the first phase from one 64-bit product per segment,
// every further one by integer additions, bit-identical to walking the program (common.hpp: the phase is exact modulo 2^64).  Samples past the span's end are zeros and
// stay zeros under a finite rotor.
struct BdRotor {
    unsigned long long ph, step, inc;
};
__device__ __forceinline__ BdRotor bd_rotor_start(const EwiseHook& h, long in0 /*float index of the segment's first staged float: even*/, int tid) {
    BdRotor r;
    r.inc  = h.rot_inc;
    r.step = 512ull * h.rot_inc;
    r.ph   = h.rot_p0 + (unsigned long long)(h.pos + 1 + in0 / 2) * h.rot_inc + (unsigned long long)(2 * tid) * h.rot_inc;
    return r;
}
__device__ __forceinline__ float4 bd_rotor_next(float4 v, BdRotor& r) { // the lane's next float4 (256 float4s further on)
#pragma clang fp contract(off)
    float c0, s0, c1, s1;
    rotor_at(r.ph, c0, s0);
    rotor_at(r.ph + r.inc, c1, s1);
    r.ph += r.step;
    return make_float4(v.x * c0 - v.y * s0, v.x * s0 + v.y * c0, v.z * c1 - v.w * s1, v.z * s1 + v.w * c1);
}
// four staged floats at stream float index fi (a multiple of 4) of the span's first segment: history in front of position 0 (as it lies: it holds what the prologue
// produced), the prologue on the samples of this span
template <bool HOOK>
__device__ __forceinline__ float4 bd_stage_slow(const float* __restrict__ x, const float* __restrict__ hist, int Kh, long n_in, long fi, const BdHooks& hk) {
    float t[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        const long i = fi + c;
        t[c]         = i >= 0 ? (i < n_in ? x[i] : 0.f) : (i >= -(long)Kh ? hist[Kh + i] : 0.f);
    }
    float4 v = make_float4(t[0], t[1], t[2], t[3]);
    if constexpr (HOOK) {
        if (hk.pre.n_ops > 0) {
            const float4 w = bd_hook4(v, hk.pre, hk.cplx, fi); // (fi, Kh and n_in are even for complex streams: a pair never straddles position 0)
            if (fi >= 0) v = w;
            else if (fi + 2 >= 0) { v.z = w.z; v.w = w.w; if (!hk.cplx && fi + 1 >= 0) v.y = w.y; }
            else if (!hk.cplx && fi + 3 >= 0) v.w = w.w;
        }
    }
    return v;
}
template <bool HOOK>
__device__ __forceinline__ void bd_new_hist(const float* __restrict__ x, const float* __restrict__ hist, int Kh, long n_in, float* __restrict__ new_hist, int tid, const BdHooks& hk) {
    if constexpr (HOOK) {
        if (hk.pre.n_ops > 0) { // Kh is a multiple of 4 floats for every filter that reaches these kernels with a prologue (fir.hip: hcap is a power of two >= 4)
            for (int h = 4 * tid; h < Kh; h += 4 * 256) {
                const float4 v = bd_stage_slow<true>(x, hist, Kh, n_in, n_in - Kh + h, hk);
                new_hist[h] = v.x; new_hist[h + 1] = v.y; new_hist[h + 2] = v.z; new_hist[h + 3] = v.w;
            }
            return;
        }
    }
    for (int h = tid; h < Kh; h += 256) {
        const long i = n_in - Kh + h;
        new_hist[h]  = i >= 0 ? x[i] : hist[Kh + i];
    }
}

} // namespace gr4
