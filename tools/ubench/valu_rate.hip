// microbenchmark: sustained VALU issue rate on gfx950 (v_fma_f32 / v_add_f32 / v_pk_fma_f32), per waves-per-SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int MODE>
__global__ void k(float* out, int iters) {
    float a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = threadIdx.x * 0.001f + i;
    const float b = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) a[i] = fmaf(a[i], b, c);
            else if (MODE == 1) a[i] = a[i] + b;
            else if (MODE == 2) a[i] = a[i] * b;
        }
        if (MODE == 3) {
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                typedef float f2 __attribute__((ext_vector_type(2)));
                f2 v = {a[i], a[i + 1]}, bb = {b, b}, cc = {c, c};
                v = __builtin_elementwise_fma(v, bb, cc);
                a[i] = v[0]; a[i + 1] = v[1];
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int MODE>
void run(const char* name, int wg_per_cu, int threads) {
    int iters = 4000, blocks = 256 * wg_per_cu;
    float* d; hipMalloc(&d, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double laneops = (double)blocks * threads * iters * 16;
    printf("%-10s waves/SIMD=%d : %.2f T lane-ops/s  (%.1f lanes/clk/SIMD @2.4GHz)\n", name, wg_per_cu * threads / 256, laneops / ms / 1e9, laneops / ms / 1e9 * 1e12 / (1024 * 2.4e9) / 1e0 / 1e0);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("fma", w, 256); run<1>("add", w, 256); run<2>("mul", w, 256); run<3>("pk_fma", w, 256);
    }
    run<0>("fma", 2, 512); run<0>("fma", 2, 1024);
}
