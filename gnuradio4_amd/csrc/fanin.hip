// fanin.hip -- the cross-device combiner edge: partial sums of the SDR-channel branches that live on different GPUs meet over RCCL (xGMI).
//
// The one exchange step of a sharded flowgraph (BASELINE.json configs[4]): math::Add<float> with n inputs (MathOpMultiPortImpl,
// blocks/math/.../Math.hpp:73-108) whose inputs sit on different "gpu:hip:i" compute domains (ComputeDomain.hpp:47-100, EdgeParameters.domain,
// BlockModel.hpp:64-72).  Every rank folds its local channels first (gr4hip_chain_process_multi / gr4hip_math_nary); what crosses the links is one
// float per output element and rank.  One process per GPU, one communicator rank per process.
//
// librccl is opened at run time (dlopen), not linked: a process that never shards never loads it, and a host that already carries an RCCL (PyTorch
// bundles its own librccl.so) keeps exactly one copy -- the loaded one is reused (RTLD_NOLOAD first).  No torch in this path.
#include "common.hpp"

#include <dlfcn.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>
#include <string>
#include <type_traits>

namespace gr4 {

// the slice of the RCCL API this file uses (rccl.h: ncclUniqueId is 128 opaque bytes, ncclFloat32 = 7, ncclSum = 0, ncclSuccess = 0)
struct RcclApi {
    using comm_t = void*;
    struct unique_id { char internal[128]; };
    int (*GetUniqueId)(unique_id*)                                                          = nullptr;
    int (*CommInitRank)(comm_t*, int, unique_id, int)                                       = nullptr;
    int (*CommDestroy)(comm_t)                                                              = nullptr;
    int (*ReduceScatter)(const void*, void*, size_t, int, int, comm_t, hipStream_t)         = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, comm_t, hipStream_t)             = nullptr;
    int (*Send)(const void*, size_t, int, int, comm_t, hipStream_t)                         = nullptr;
    int (*Recv)(void*, size_t, int, int, comm_t, hipStream_t)                               = nullptr;
    int (*GroupStart)()                                                                     = nullptr;
    int (*GroupEnd)()                                                                       = nullptr;
    const char* (*GetErrorString)(int)                                                      = nullptr;
    void*       lib = nullptr;
    std::string why;
};
constexpr int kNcclFloat = 7, kNcclSum = 0;

static RcclApi& rccl() {
    static RcclApi       api;
    static std::once_flag once;
    std::call_once(once, [] {
        const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
        for (const char* n : names) // an RCCL the process already carries wins: one copy of the library, one set of its globals
            if ((api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL)) != nullptr) break;
        if (const char* e = std::getenv("GR4HIP_RCCL_LIBRARY"); !api.lib && e) api.lib = dlopen(e, RTLD_NOW | RTLD_GLOBAL);
        for (const char* n : names) {
            if (api.lib) break;
            api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
        }
        if (!api.lib) {
            const char* e = dlerror(); // (once: the call clears the error it returns)
            api.why = std::string("librccl not found (") + (e ? e : "?") + "); set GR4HIP_RCCL_LIBRARY";
            return;
        }
        auto get = [&](const char* sym, auto& fn) {
            fn = reinterpret_cast<std::remove_reference_t<decltype(fn)>>(dlsym(api.lib, sym));
            if (!fn && api.why.empty()) api.why = std::string("librccl lacks ") + sym;
        };
        get("ncclGetUniqueId", api.GetUniqueId);
        get("ncclCommInitRank", api.CommInitRank);
        get("ncclCommDestroy", api.CommDestroy);
        get("ncclReduceScatter", api.ReduceScatter);
        get("ncclAllReduce", api.AllReduce);
        get("ncclSend", api.Send);
        get("ncclRecv", api.Recv);
        get("ncclGroupStart", api.GroupStart);
        get("ncclGroupEnd", api.GroupEnd);
        get("ncclGetErrorString", api.GetErrorString);
    });
    return api;
}

#define GR4_RCCL_TRY(expr)                                                                                  \
    do {                                                                                                    \
        const int e_ = (expr);                                                                              \
        if (e_ != 0) {                                                                                      \
            ::gr4::set_error("%s failed: %s", #expr, api.GetErrorString ? api.GetErrorString(e_) : "?");    \
            return GR4HIP_RUNTIME_ERROR;                                                                    \
        }                                                                                                   \
    } while (0)

// out[i] = ((in[0][i] + in[1][i]) + ...) over the n slabs of an all-to-all landing area: math::Add's left fold in RANK order, the same on every rank
__global__ __launch_bounds__(256) void fanin_fold_kernel(const float4* __restrict__ recv, float4* __restrict__ out, long n4, int n_ranks) {
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
        float4 a = recv[i];
        for (int r = 1; r < n_ranks; ++r) {
            const float4 b = recv[(long)r * n4 + i];
            a = make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
        }
        out[i] = a;
    }
}

} // namespace gr4

using namespace gr4;

struct gr4hip_fanin {
    RcclApi::comm_t comm = nullptr;
    int             rank = 0, n_ranks = 1, device = 0;
};

extern "C" {

int gr4hip_fanin_unique_id(void* id128) {
    GR4_REQUIRE(id128, "fanin_unique_id: null buffer");
    RcclApi& api = rccl();
    if (!api.why.empty()) { set_error("fan-in: %s", api.why.c_str()); return GR4HIP_UNSUPPORTED; }
    RcclApi::unique_id id;
    GR4_RCCL_TRY(api.GetUniqueId(&id));
    std::memcpy(id128, id.internal, sizeof(id.internal));
    return GR4HIP_OK;
}

int gr4hip_fanin_create(gr4hip_fanin_t** out, const void* id128, int rank, int n_ranks) {
    GR4_REQUIRE(out && id128, "fanin_create: null argument");
    GR4_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "fanin_create: rank %d of %d", rank, n_ranks);
    RcclApi& api = rccl();
    if (!api.why.empty()) { set_error("fan-in: %s", api.why.c_str()); return GR4HIP_UNSUPPORTED; }
    auto* f = new (std::nothrow) gr4hip_fanin();
    GR4_REQUIRE(f, "out of host memory");
    f->rank = rank;
    f->n_ranks = n_ranks;
    if (hipGetDevice(&f->device) != hipSuccess) { delete f; set_error("fanin_create: no current device"); return GR4HIP_NO_DEVICE; }
    RcclApi::unique_id id;
    std::memcpy(id.internal, id128, sizeof(id.internal));
    const int e = api.CommInitRank(&f->comm, n_ranks, id, rank); // collective: returns when every rank has called it
    if (e != 0) {
        set_error("ncclCommInitRank failed: %s", api.GetErrorString ? api.GetErrorString(e) : "?");
        delete f;
        return GR4HIP_RUNTIME_ERROR;
    }
    *out = f;
    return GR4HIP_OK;
}

int gr4hip_fanin_rank(const gr4hip_fanin_t* f, int* rank, int* n_ranks) {
    GR4_REQUIRE(f, "fanin_rank: null handle");
    if (rank) *rank = f->rank;
    if (n_ranks) *n_ranks = f->n_ranks;
    return GR4HIP_OK;
}

int gr4hip_fanin_reduce_scatter_sum_f32(gr4hip_fanin_t* f, const float* d_partial, float* d_shard, size_t shard_count, gr4hip_stream_t stream) {
    GR4_REQUIRE(f && d_partial && d_shard, "fanin_reduce_scatter: null argument");
    if (shard_count == 0) return GR4HIP_OK;
    RcclApi& api = rccl();
    GR4_RCCL_TRY(api.ReduceScatter(d_partial, d_shard, shard_count, kNcclFloat, kNcclSum, f->comm, as_stream(stream)));
    return GR4HIP_OK;
}

int gr4hip_fanin_all_reduce_sum_f32(gr4hip_fanin_t* f, const float* d_partial, float* d_sum, size_t count, gr4hip_stream_t stream) {
    GR4_REQUIRE(f && d_partial && d_sum, "fanin_all_reduce: null argument");
    if (count == 0) return GR4HIP_OK;
    RcclApi& api = rccl();
    GR4_RCCL_TRY(api.AllReduce(d_partial, d_sum, count, kNcclFloat, kNcclSum, f->comm, as_stream(stream)));
    return GR4HIP_OK;
}

int gr4hip_fanin_all_to_all_sum_f32(gr4hip_fanin_t* f, const float* d_partial, float* d_scratch, float* d_shard, size_t shard_count, gr4hip_stream_t stream) {
    GR4_REQUIRE(f && d_partial && d_scratch && d_shard, "fanin_all_to_all: null argument");
    GR4_REQUIRE(shard_count % 4 == 0 && (reinterpret_cast<uintptr_t>(d_scratch) | reinterpret_cast<uintptr_t>(d_shard)) % 16 == 0, "fanin_all_to_all: 16-byte-aligned shards of a multiple of 4 floats");
    if (shard_count == 0) return GR4HIP_OK;
    RcclApi&    api = rccl();
    hipStream_t st  = as_stream(stream);
    GR4_RCCL_TRY(api.GroupStart()); // shard j of this rank's partial sum goes straight to rank j: one xGMI link per peer, all of them busy at once
    for (int r = 0; r < f->n_ranks; ++r) {
        const int e1 = api.Send(d_partial + (size_t)r * shard_count, shard_count, kNcclFloat, r, f->comm, st);
        const int e2 = e1 ? 0 : api.Recv(d_scratch + (size_t)r * shard_count, shard_count, kNcclFloat, r, f->comm, st);
        if (e1 || e2) { (void)api.GroupEnd(); set_error("ncclSend / ncclRecv failed: %s", api.GetErrorString ? api.GetErrorString(e1 ? e1 : e2) : "?"); return GR4HIP_RUNTIME_ERROR; }
    }
    GR4_RCCL_TRY(api.GroupEnd());
    const long n4 = (long)(shard_count / 4);
    hipLaunchKernelGGL(fanin_fold_kernel, dim3((unsigned)std::min<long>(ceil_div(n4, 256L), 4096)), dim3(256), 0, st, reinterpret_cast<const float4*>(d_scratch),
                       reinterpret_cast<float4*>(d_shard), n4, f->n_ranks);
    GR4_LAUNCH_CHECK();
    return GR4HIP_OK;
}

int gr4hip_fanin_destroy(gr4hip_fanin_t* f) {
    if (!f) return GR4HIP_OK;
    RcclApi& api = rccl();
    if (f->comm && api.CommDestroy) (void)api.CommDestroy(f->comm);
    delete f;
    return GR4HIP_OK;
}

} // extern "C"
