// developer probe: accuracy of v_sin_f32 / v_cos_f32 (argument in turns) on [-0.5, 0.5] against float64, and of a quadrant-folded polynomial pair
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/native_sincos_accuracy.hip -o gpurun_out/native_sincos
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
__device__ __forceinline__ void poly_sincos_turns(float t, float& sn, float& cs) { // t in [-0.5, 0.5] turns
    const float q = rintf(4.f * t);                 // quadrant -2 .. 2
    const float r = fmaf(q, -0.25f, t);             // [-1/8, 1/8] turns (exact)
    const float a = r * 6.283185307179586f, a2 = a * a;
    float s = fmaf(a2, fmaf(a2, fmaf(a2, -1.9515295891e-4f, 8.3321608736e-3f), -1.6666654611e-1f), 1.f) * a;
    float c = fmaf(a2, fmaf(a2, fmaf(a2, fmaf(a2, 2.443315711809948e-5f, -1.388731625493765e-3f), 4.166664568298827e-2f), -0.5f), 1.f);
    const int qi = (int)q & 3;
    const float s2 = (qi & 1) ? c : s, c2 = (qi & 1) ? s : c;
    sn = (qi == 2 || qi == 3) ? -s2 : s2;           // q=1: sin = c, cos = -s; q=2: sin = -s, cos = -c; q=3: sin = -c, cos = s
    cs = (qi == 1 || qi == 2) ? -c2 : c2;
}
__global__ void probe(double* out, int n) {
    double e_nat = 0, e_lib = 0, e_pol = 0;
    for (long i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const double t = -0.5 + (double)i / (double)n;
        const float  tf = (float)t;
        const double s = sin(6.283185307179586476925286766559 * (double)tf), c = cos(6.283185307179586476925286766559 * (double)tf);
        const float sn = __builtin_amdgcn_sinf(tf), cn = __builtin_amdgcn_cosf(tf);
        float sl, cl;
        sincosf((float)((double)tf * 6.283185307179586476925286766559), &sl, &cl);
        float sp, cp;
        poly_sincos_turns(tf, sp, cp);
        e_nat = fmax(e_nat, fmax(fabs(sn - s), fabs(cn - c)));
        e_lib = fmax(e_lib, fmax(fabs(sl - s), fabs(cl - c)));
        e_pol = fmax(e_pol, fmax(fabs(sp - s), fabs(cp - c)));
    }
    atomicMax((unsigned long long*)out, __double_as_longlong(e_nat));
    atomicMax((unsigned long long*)out + 1, __double_as_longlong(e_lib));
    atomicMax((unsigned long long*)out + 2, __double_as_longlong(e_pol));
}
int main() {
    double* d; hipMalloc(&d, 24); hipMemset(d, 0, 24);
    probe<<<1024, 256>>>(d, 1 << 28);
    double h[3]; hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
    printf("max abs error over 2^28 arguments in [-0.5, 0.5) turns: v_sin/v_cos %.3e   sincosf(float(2 pi t)) %.3e   folded polynomial %.3e\n", h[0], h[1], h[2]);
    return 0;
}
