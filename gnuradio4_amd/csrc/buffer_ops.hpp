// buffer_ops.hpp -- SRSRC (buffer) loads and stores for gfx950
#pragma once
#include <hip/hip_runtime.h>

namespace gr4 {

// buffer (SRSRC) accesses: wave-uniform descriptor + one 32-bit lane offset + a scalar offset per access, so no per-access
// 64-bit address lives in VGPRs (with flat addressing hipcc hoists 40+ lane addresses out of the frame loop and spills them)
#ifndef GR4_BUF_LOAD_AUX // developer builds: 2 = nt (streaming) on the sample loads -- profiles/r05_streaming_hints.txt
#define GR4_BUF_LOAD_AUX 0
#endif
using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void* p, unsigned bytes) { return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000); }
__device__ __forceinline__ float2 buf_load_f2(rsrc_t r, int voff, int soff) {
    const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, GR4_BUF_LOAD_AUX);
    return make_float2(__uint_as_float(v[0]), __uint_as_float(v[1]));
}
#ifndef GR4_BUF_STORE_AUX // developer builds: 2 = nt (streaming) on the result stores -- profiles/r05_headline_bounds.txt
#define GR4_BUF_STORE_AUX 0
#endif
__device__ __forceinline__ void buf_store_f(rsrc_t r, float v, int voff, int soff) { __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(v), r, voff, soff, GR4_BUF_STORE_AUX); }
__device__ __forceinline__ void buf_store_f2(rsrc_t r, float2 v, int voff, int soff) {
    using u32x2 = __attribute__((ext_vector_type(2))) unsigned;
    u32x2 d = {__float_as_uint(v.x), __float_as_uint(v.y)};
    __builtin_amdgcn_raw_buffer_store_b64(d, r, voff, soff, GR4_BUF_STORE_AUX);
}
__device__ __forceinline__ float buf_load_f(rsrc_t r, int voff, int soff) { return __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, GR4_BUF_LOAD_AUX)); }

} // namespace gr4
