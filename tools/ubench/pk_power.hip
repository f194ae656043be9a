// microbenchmark: scalar vs packed f32 complex arithmetic at the package power cap -- which form does more work per joule?
// usage: pk_power <mode> <seconds>   modes: 2 = 2 x v_fma_f32, 3 = v_pk_fma_f32, 5 = complex multiply scalar (2 mul + 2 fma), 4 = packed (pk_mul + pk_fma)
// run beside `rocm-smi --showpower --showclocks` (tools/pk_power.sh); build with -fno-slp-vectorize
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters) {
    f2 a[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) a[i] = f2{threadIdx.x * 0.001f + i, 1.0f + i * 0.37f};
    const f2 b = {0.99990001f, 0.01414f}, c = {1e-4f, -2e-4f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 2) { a[i].x = fmaf(a[i].x, b.x, c.x); a[i].y = fmaf(a[i].y, b.y, c.y); }
            else if (MODE == 3) a[i] = __builtin_elementwise_fma(a[i], b, c);
            else if (MODE == 4) { f2 t, r; const f2 x = a[i];
                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(x), "v"(b));
                asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(x), "v"(b), "v"(t));
                a[i] = r; }
            else { const f2 x = a[i]; a[i].x = fmaf(x.x, b.x, -x.y * b.y); a[i].y = fmaf(x.x, b.y, x.y * b.x); }
        }
    }
    f2 s = a[0];
#pragma unroll
    for (int i = 1; i < 16; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
template <int MODE>
double launch(float* d, int iters) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256 * 4), dim3(256), 0, 0, d, iters); // 4 workgroups of 4 waves per CU = 4 waves per SIMD
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e-3;
}
int main(int argc, char** argv) {
    const int mode = argc > 1 ? atoi(argv[1]) : 2;
    const double secs = argc > 2 ? atof(argv[2]) : 3.0;
    float* d; hipMalloc(&d, (size_t)256 * 4 * 256 * 4);
    auto run = [&](int iters) { return mode == 2 ? launch<2>(d, iters) : mode == 3 ? launch<3>(d, iters) : mode == 4 ? launch<4>(d, iters) : launch<5>(d, iters); };
    run(1000);
    const int iters = 200000;
    double t = 0, work = 0;
    while (t < secs) { const double dt = run(iters); t += dt; work += (double)256 * 4 * 256 * 16 * iters; } // complex elements updated
    const char* names[] = {"", "", "2 x v_fma_f32", "v_pk_fma_f32", "cmul packed (pk_mul + pk_fma)", "cmul scalar (2 mul + 2 fma)"};
    printf("%-32s %.2f T element-updates/s over %.1f s\n", names[mode], work / t / 1e12, t);
    return 0;
}
