// stream_rate.hip -- out[i] = in[i] * c over 2^28 int32 (1 GiB in, 1 GiB out): which launch shape / cache policy streams fastest on gfx950?
// hipcc --offload-arch=gfx950 -O3 tools/ubench/stream_rate.hip -o /tmp/stream_rate && /tmp/stream_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned int u4 __attribute__((ext_vector_type(4)));
#define uint4 u4
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

template <int NT, int UNROLL>
__global__ void k_stride(const uint4* __restrict__ in, uint4* __restrict__ out, long nvec, int c) {
    const long stride = (long)gridDim.x * blockDim.x;
    long v = (long)blockIdx.x * blockDim.x + threadIdx.x;
    for (; v + (UNROLL - 1) * stride < nvec; v += UNROLL * stride) {
        uint4 a[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) a[u] = NT ? __builtin_nontemporal_load(&in[v + u * stride]) : in[v + u * stride];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            a[u].x *= c; a[u].y *= c; a[u].z *= c; a[u].w *= c;
            if (NT) __builtin_nontemporal_store(a[u], &out[v + u * stride]); else out[v + u * stride] = a[u];
        }
    }
    for (; v < nvec; v += stride) { uint4 a = in[v]; a.x *= c; a.y *= c; a.z *= c; a.w *= c; out[v] = a; }
}
// each workgroup owns a contiguous slab
template <int NT>
__global__ void k_slab(const uint4* __restrict__ in, uint4* __restrict__ out, long nvec, int c) {
    const long per = (nvec + gridDim.x - 1) / gridDim.x, lo = per * blockIdx.x, hi = lo + per < nvec ? lo + per : nvec;
    for (long v = lo + threadIdx.x; v < hi; v += blockDim.x) {
        uint4 a = NT ? __builtin_nontemporal_load(&in[v]) : in[v];
        a.x *= c; a.y *= c; a.z *= c; a.w *= c;
        if (NT) __builtin_nontemporal_store(a, &out[v]); else out[v] = a;
    }
}
int main() {
    const long n = 1L << 28, nvec = n / 4;
    uint4 *in, *out;
    CK(hipMalloc(&in, n * 4)); CK(hipMalloc(&out, n * 4));
    CK(hipMemset(in, 1, n * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    auto run = [&](const char* name, auto launch) {
        for (int i = 0; i < 5; ++i) launch();
        hipEventRecord(e0);
        for (int i = 0; i < 20; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-44s %7.1f Gsamples/s  %5.2f TB/s\n", name, n * 20 / (ms * 1e-3) / 1e9, n * 8.0 * 20 / (ms * 1e-3) / 1e12);
    };
    for (int grid : {1024, 2048, 4096, 8192, 65536}) {
        char nm[96];
        snprintf(nm, 96, "grid-stride 256 thr, grid %d", grid);                    run(nm, [&] { k_stride<0, 1><<<grid, 256>>>(in, out, nvec, 3); });
        snprintf(nm, 96, "grid-stride 256 thr, grid %d, nt", grid);                run(nm, [&] { k_stride<1, 1><<<grid, 256>>>(in, out, nvec, 3); });
        snprintf(nm, 96, "grid-stride 256 thr, grid %d, unroll 2", grid);          run(nm, [&] { k_stride<0, 2><<<grid, 256>>>(in, out, nvec, 3); });
        snprintf(nm, 96, "grid-stride 256 thr, grid %d, unroll 4, nt", grid);      run(nm, [&] { k_stride<1, 4><<<grid, 256>>>(in, out, nvec, 3); });
    }
    for (int grid : {2048, 8192, 65536}) {
        char nm[96];
        snprintf(nm, 96, "slab per WG, grid %d", grid);     run(nm, [&] { k_slab<0><<<grid, 256>>>(in, out, nvec, 3); });
        snprintf(nm, 96, "slab per WG, grid %d, nt", grid); run(nm, [&] { k_slab<1><<<grid, 256>>>(in, out, nvec, 3); });
    }
    run("hipMemcpyDtoD", [&] { hipMemcpyAsync(out, in, n * 4, hipMemcpyDeviceToDevice, 0); });
    return 0;
}
