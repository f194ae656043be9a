#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned* out) {
    unsigned a = threadIdx.x, b = 1000 + threadIdx.x;
    auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
    out[threadIdx.x * 4 + 0] = r[0];
    out[threadIdx.x * 4 + 1] = r[1];
    auto s = __builtin_amdgcn_permlane16_swap(a, b, false, false);
    out[threadIdx.x * 4 + 2] = s[0];
    out[threadIdx.x * 4 + 3] = s[1];
}
// throughput: chain of swaps + fmas
template <int MODE>
__global__ void rate(float* out, int iters) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = threadIdx.x * 0.001f + i;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 1) {
                auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 8]), false, false);
                v[i] = __uint_as_float(r[0]); v[i + 8] = __uint_as_float(r[1]);
            } else if (MODE == 2) {
                auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v[i]), __float_as_uint(v[i + 8]), false, false);
                v[i] = __uint_as_float(r[0]); v[i + 8] = __uint_as_float(r[1]);
            } else if (MODE == 3) { // dpp mov pair
                float p = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i]), 0xB1, 0xF, 0xF, false));
                float q = __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v[i + 8]), 0xB1, 0xF, 0xF, false));
                v[i] = p; v[i + 8] = q;
            }
            v[i] = fmaf(v[i], 1.0001f, 0.5f); v[i + 8] = fmaf(v[i + 8], 0.9999f, 0.25f);
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE> void run(const char* n) {
    int iters = 2000, blocks = 256, threads = 1024;
    float* d; hipMalloc(&d, blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 10);
    hipEventRecord(e0); hipLaunchKernelGGL(rate<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-12s %.3f ms  -> %.2f ns per inner group (16 fma [+8 exch]) per wave-slot\n", n, ms, ms * 1e6 / iters / 8.0);
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 16); k<<<1, 64>>>(d); unsigned h[256]; hipMemcpy(h, d, 1024, hipMemcpyDeviceToHost);
    for (int l : {0, 1, 15, 16, 17, 31, 32, 33, 47, 48, 63}) printf("lane %2d: swap32 -> (%u, %u)  swap16 -> (%u, %u)\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    run<0>("fma only"); run<1>("swap32"); run<2>("swap16"); run<3>("2x dpp mov");
}
