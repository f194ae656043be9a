// microbenchmark: cycles per wave-instruction of scalar vs packed f32 VALU ops on gfx950, per waves-per-SIMD (build with -fno-slp-vectorize)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void k(float* out, int iters) {
    f2 a[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) a[i] = f2{threadIdx.x * 0.001f + i, 1.0f + i};
    const f2 b = {1.0001f, 0.9999f}, c = {0.5f, 0.25f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) { a[i].x = a[i].x + b.x; a[i].y = a[i].y + b.y; }            // 2 v_add_f32
            else if (MODE == 1) a[i] = a[i] + b;                                           // 1 v_pk_add_f32
            else if (MODE == 2) { a[i].x = fmaf(a[i].x, b.x, c.x); a[i].y = fmaf(a[i].y, b.y, c.y); } // 2 v_fma
            else if (MODE == 3) a[i] = __builtin_elementwise_fma(a[i], b, c);              // 1 v_pk_fma_f32
            else if (MODE == 4) { f2 t, r; const f2 x = a[i];                              // complex multiply, packed (2 instructions)
                asm("v_pk_mul_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[1,0] neg_lo:[0,1]" : "=v"(t) : "v"(x), "v"(b));
                asm("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[0,1,1]" : "=v"(r) : "v"(x), "v"(b), "v"(t));
                a[i] = r; }
            else { const f2 x = a[i];                                                       // complex multiply, scalar (2 mul + 2 fma)
                a[i].x = fmaf(x.x, b.x, -x.y * b.y); a[i].y = fmaf(x.x, b.y, x.y * b.x); }
        }
    }
    f2 s = a[0];
#pragma unroll
    for (int i = 1; i < 8; ++i) s += a[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s.x + s.y;
}
template <int MODE>
void run(const char* name, int wps, int instr_per_elem) {
    const int iters = 4000, threads = 256, blocks = 256 * wps;
    float* d; hipMalloc(&d, (size_t)blocks * threads * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, 10);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(threads), 0, 0, d, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    // wave-instructions per SIMD: (blocks*threads/64 waves) * iters * 8 * instr / 1024 SIMDs
    const double winstr = (double)blocks * threads / 64 * iters * 8 * instr_per_elem / 1024.0;
    printf("%-22s waves/SIMD=%d : %6.2f ns per 8 elements per wave, %.2f cycles per wave-instruction @2.1GHz (SIMD shared by %d waves)\n", name, wps,
           ms * 1e6 / iters / (blocks * threads / 64 / 1024.0), ms * 1e-3 * 2.1e9 / winstr, wps);
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4}) {
        run<0>("2 x v_add_f32", w, 2); run<1>("v_pk_add_f32", w, 1); run<2>("2 x v_fma_f32", w, 2); run<3>("v_pk_fma_f32", w, 1);
        run<5>("cmul scalar (2mul+2fma)", w, 4); run<4>("cmul packed (pk_mul+pk_fma)", w, 2);
    }
}
