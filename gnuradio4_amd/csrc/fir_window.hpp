// fir_window.hpp -- register sliding-window FIR step shared by fir.hip and chain_fused.hip (gfx950).
#pragma once
#include <hip/hip_runtime.h>

namespace gr4 {

constexpr int kFirR = 8; // output floats per lane

template <int S, int ROT>
__device__ __forceinline__ void fir_step(float (&acc)[kFirR], float (&w)[12], const float* __restrict__ tg, const float* __restrict__ xnext, bool load_next) {
    constexpr int E = 4 / S; // taps consumed per 4-float window advance
    float         tb[E];
    if constexpr (S == 1) {
        const float4 t4 = *reinterpret_cast<const float4*>(tg);
        tb[0] = t4.x; tb[1] = t4.y; tb[2] = t4.z; tb[3] = t4.w;
    } else {
        const float2 t2 = *reinterpret_cast<const float2*>(tg);
        tb[0] = t2.x; tb[1] = t2.y;
    }
#pragma unroll
    for (int e = 0; e < E; ++e)
#pragma unroll
        for (int r = 0; r < kFirR; ++r) acc[r] = fmaf(tb[e], w[((4 + r - S * e) + 4 * ROT) % 12], acc[r]);
    if (load_next) { // the chunk that just left the window (logical 8..11) is refilled with the next lower chunk
        const float4 n4 = *reinterpret_cast<const float4*>(xnext);
        w[(8 + 4 * ROT) % 12 + 0] = n4.x;
        w[(8 + 4 * ROT) % 12 + 1] = n4.y;
        w[(8 + 4 * ROT) % 12 + 2] = n4.z;
        w[(8 + 4 * ROT) % 12 + 3] = n4.w;
    }
}

} // namespace gr4
