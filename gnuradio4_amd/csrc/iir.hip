// iir.hip -- IIR section cascades on gfx950: exact parallel-in-time evaluation of a sequential recurrence.
//
// Replaces gr::filter::Filter<float>::processOne (std::accumulate over sections of detail::computeFilter,
// algorithm/.../filter/FilterTool.hpp:116-158, 244-246) and gr::filter::iir_filter<float,form>::processOne
// (blocks/filter/.../time_domain_filter.hpp:89-121).  The cascade is one LTI system with state s (M floats, the
// direct-form-II delay lines of all sections).  For a chunk of L samples:  s_end = Phi_L * s_start + z  with z the
// zero-state response end state, so chunk start states follow from a (matrix) prefix scan:
//   pass Z : every lane runs its L-sample chunk from zero state (tile staged in LDS, conflict-free rows) -> z_c;
//            in-block Kogge-Stone scan with host-precomputed Phi_{L*2^k} gives the block's zero-state end state;
//   pass B : one wave chains the block states  T_{b+1} = Phi_B * T_b + Z_b  (the only sequential step: n/8192 steps);
//   pass Y : same in-block scan seeded with T_b gives every chunk's true start state; the chunk is re-run from it,
//            results go back to the LDS tile and leave with coalesced stores.
// Bound: FP32 FMA latency chains, 2 recurrence passes per sample (DESIGN.md "iir_cascade").
#include "common.hpp"

#include <map>
#include <mutex>

#include <cmath>
#include <cstdlib>

namespace gr4 {

constexpr int kIirL       = 32;  // samples per lane chunk
constexpr int kIirBS      = 256; // lanes per block
constexpr int kIirMaxM    = 16;  // total state floats
constexpr int kIirRounds  = 8;   // log2(kIirBS)
constexpr int kIirBB      = 512; // blocks per group of the block-level scan
constexpr int kIirBRounds = 9;   // log2(kIirBB)
constexpr int kIirBShift  = 2;   // log2(kIirBK)
constexpr int kIirBK      = 4;   // blocks chained per lane in the block-level scan

// Kernels are instantiated for (section order, padded section count): the state count MP = ORD * NSEC is a compile-time 4, 8 or 16,
// so the propagation matrices come in through wide scalar loads and everything unrolls.  A cascade with fewer sections is padded
// with identity sections (b0 = 1): y = x exactly, no per-section branches in the sample loop.
template <int ORD, int NSEC>
struct IirCoef {
    float b[NSEC][ORD + 1];
    float a[NSEC][ORD + 1]; // a[.][0] unused (== 1)
};

// one sample through the cascade, direct form II per section (FilterTool.hpp:130-141)
template <int ORD, int NSEC>
__device__ __forceinline__ float iir_step(const IirCoef<ORD, NSEC>& c, float (&st)[NSEC][ORD], float x) {
#pragma unroll
    for (int s = 0; s < NSEC; ++s) {
        float w = x;
#pragma unroll
        for (int j = 0; j < ORD; ++j) w = fmaf(-c.a[s][j + 1], st[s][j], w);
        float y = c.b[s][0] * w;
#pragma unroll
        for (int j = 0; j < ORD; ++j) y = fmaf(c.b[s][j + 1], st[s][j], y);
#pragma unroll
        for (int j = ORD - 1; j > 0; --j) st[s][j] = st[s][j - 1];
        st[s][0] = w;
        x        = y;
    }
    return x;
}

// propagation matrices global -> LDS, transposed ([round][j][i]) so that the scan reads a column of Phi as MP/4 broadcast b128 loads.
// (Scalar loads straight from global memory cost a cold ~1 us round trip in every scan round of every block.)
template <int MP, int ROUNDS>
__device__ __forceinline__ void iir_load_phi(float* pl, const float* __restrict__ phi) {
    for (int e = threadIdx.x; e < ROUNDS * MP * MP; e += blockDim.x) {
        const int k = e / (MP * MP), r = e % (MP * MP), i = r / MP, j = r % MP;
        pl[(k * MP + j) * MP + i] = phi[e];
    }
}

// inclusive scan over the block's chunks:  sv[.][c] <- sum_{i<=c} Phi^{c-i} sv[.][i]   (sv is [MP][BS]: lane c owns column c, so every
// access is lane-contiguous -- a [BS][MP] layout puts 16 lanes on one bank)
template <int MP, int BS, int ROUNDS>
__device__ __forceinline__ void iir_block_scan(float* sv, const float* pl /*LDS [ROUNDS][MP(j)][MP(i)]: Phi^(2^k) transposed*/) {
    const int c = threadIdx.x;
#pragma unroll 1
    for (int k = 0; k < ROUNDS; ++k) {
        const int    off = 1 << k;
        float        tmp[MP];
        const float* P = pl + k * MP * MP;
        if (c >= off) {
#pragma unroll
            for (int i = 0; i < MP; ++i) tmp[i] = 0.f;
#pragma unroll
            for (int j = 0; j < MP; ++j) {
                const float sj = sv[j * BS + c - off];
#pragma unroll
                for (int i = 0; i < MP; ++i) tmp[i] = fmaf(P[j * MP + i], sj, tmp[i]);
            }
        }
        __syncthreads();
        if (c >= off) {
#pragma unroll
            for (int i = 0; i < MP; ++i) sv[i * BS + c] += tmp[i];
        }
        __syncthreads();
    }
}

// HBM -> LDS tile [chunk][L + 1]: all eight 16-byte loads of a lane are in flight before the first LDS write (a load-per-iteration
// loop pays the full memory latency 32 times per block)
using iir_f32x4 = __attribute__((ext_vector_type(4))) float;
// nt (uniform): streaming hints on the tile's loads / stores -- a span that cannot stay in the 256 MB of memory-side cache anyway (profiles/r05_streaming_hints.txt: 4 biquads
// on 2^26 / 2^27 samples + 3.5 %, 1-pole + 3 / + 6 %; on 2^24 samples, which a back-to-back loop re-reads from that cache, - 6 %: the host sets it from the span's size)
__device__ __forceinline__ void iir_stage_tile(float* tile, const float* __restrict__ x, long base, long n, bool nt = false) {
    constexpr int kV = kIirL / 4; // float4 per lane
    if (base + (long)kIirBS * kIirL <= n && (reinterpret_cast<uintptr_t>(x + base) & 15) == 0) {
        float4 v[kV];
        if (nt) {
#pragma unroll
            for (int q = 0; q < kV; ++q) { const iir_f32x4 t = __builtin_nontemporal_load(reinterpret_cast<const iir_f32x4*>(x + base) + (q * kIirBS + threadIdx.x)); v[q] = make_float4(t[0], t[1], t[2], t[3]); }
        } else {
#pragma unroll
            for (int q = 0; q < kV; ++q) v[q] = reinterpret_cast<const float4*>(x + base)[q * kIirBS + threadIdx.x];
        }
#pragma unroll
        for (int q = 0; q < kV; ++q) {
            const int s = 4 * (q * kIirBS + threadIdx.x);
            float*    d = tile + (s / kIirL) * (kIirL + 1) + (s % kIirL);
            d[0] = v[q].x; d[1] = v[q].y; d[2] = v[q].z; d[3] = v[q].w;
        }
    } else {
        for (int s = threadIdx.x; s < kIirBS * kIirL; s += kIirBS) {
            const long i = base + s;
            tile[(s / kIirL) * (kIirL + 1) + (s % kIirL)] = i < n ? x[i] : 0.f;
        }
    }
}
__device__ __forceinline__ void iir_unstage_tile(const float* tile, float* __restrict__ y, long base, long n, bool nt = false) {
    constexpr int kV = kIirL / 4;
    if (base + (long)kIirBS * kIirL <= n && (reinterpret_cast<uintptr_t>(y + base) & 15) == 0) {
#pragma unroll
        for (int q = 0; q < kV; ++q) {
            const int    s = 4 * (q * kIirBS + threadIdx.x);
            const float* d = tile + (s / kIirL) * (kIirL + 1) + (s % kIirL);
            if (nt) __builtin_nontemporal_store(iir_f32x4{d[0], d[1], d[2], d[3]}, reinterpret_cast<iir_f32x4*>(y + base) + (q * kIirBS + threadIdx.x));
            else reinterpret_cast<float4*>(y + base)[q * kIirBS + threadIdx.x] = make_float4(d[0], d[1], d[2], d[3]);
        }
    } else {
        for (int s = threadIdx.x; s < kIirBS * kIirL; s += kIirBS) {
            const long i = base + s;
            if (i < n) y[i] = tile[(s / kIirL) * (kIirL + 1) + (s % kIirL)];
        }
    }
}

// pass Z: z_c per chunk (global, [chunks][MP]) and the block's zero-state end state Zb[block][MP]
template <int ORD, int NSEC>
__global__ __launch_bounds__(kIirBS) void iir_pass_z(const float* __restrict__ x, long n, IirCoef<ORD, NSEC> coef, const float* __restrict__ phi, float* __restrict__ zc,
                                                      float* __restrict__ zb) {
    constexpr int    MP = ORD * NSEC;
    __shared__ float tile[kIirBS * (kIirL + 1)];
    __shared__ float sv[MP * kIirBS];
    __shared__ __attribute__((aligned(16))) float pl[kIirRounds * MP * MP];
    const long       base = (long)blockIdx.x * kIirBS * kIirL;
    iir_load_phi<MP, kIirRounds>(pl, phi);
    iir_stage_tile(tile, x, base, n);
    __syncthreads();
    float st[NSEC][ORD];
#pragma unroll
    for (int s = 0; s < NSEC; ++s)
#pragma unroll
        for (int j = 0; j < ORD; ++j) st[s][j] = 0.f;
    const float* row = tile + threadIdx.x * (kIirL + 1);
#pragma unroll 4
    for (int i = 0; i < kIirL; ++i) (void)iir_step<ORD, NSEC>(coef, st, row[i]);
    const long chunk = (long)blockIdx.x * kIirBS + threadIdx.x;
#pragma unroll
    for (int s = 0; s < NSEC; ++s)
#pragma unroll
        for (int j = 0; j < ORD; ++j) {
            sv[(s * ORD + j) * kIirBS + threadIdx.x] = st[s][j];
            zc[chunk * MP + s * ORD + j]              = st[s][j];
        }
    __syncthreads();
    iir_block_scan<MP, kIirBS, kIirRounds>(sv, pl);
    if (threadIdx.x < MP) zb[(long)blockIdx.x * MP + threadIdx.x] = sv[threadIdx.x * kIirBS + kIirBS - 1];
}

// pass B: T_0 = carried state; T_{b+1} = Phi_B T_b + Zb[b].  Writes T_b (state at the START of block b).  The same affine scan one
// level up: every lane chains kIirBK consecutive blocks locally, the workgroup scans the lanes (Kogge-Stone with Phi_B^(kIirBK 2^k)),
// and only groups of kIirBB * kIirBK blocks (16 M samples) are chained sequentially.
template <int MP>
__global__ __launch_bounds__(kIirBB) void iir_pass_b(const float* __restrict__ state0, const float* __restrict__ phiB /*[kIirBShift + kIirBRounds][MP][MP]: Phi_B^(2^k)*/,
                                                      const float* __restrict__ zb, float* __restrict__ tb, long nblocks) {
    __shared__ float sv[MP * kIirBB];
    __shared__ float T[MP];
    __shared__ __attribute__((aligned(16))) float pl[(kIirBShift + kIirBRounds) * MP * MP];
    const int        c = threadIdx.x;
    iir_load_phi<MP, kIirBShift + kIirBRounds>(pl, phiB);
    if (c < MP) T[c] = state0[c];
    __syncthreads();
    const float* PB = pl;                          // Phi_B (transposed)
    const float* PK = pl + kIirBShift * MP * MP;   // Phi_B^kIirBK and its 2^k powers
    for (long g0 = 0; g0 < nblocks; g0 += (long)kIirBB * kIirBK) {
        const long b0 = g0 + (long)c * kIirBK;
        float      e[MP];
        // one block step of the lane's chain: e <- Phi_B e + z
        auto step = [&](const float (&z)[MP]) {
            float nv[MP];
#pragma unroll
            for (int i = 0; i < MP; ++i) nv[i] = z[i];
#pragma unroll
            for (int j = 0; j < MP; ++j)
#pragma unroll
                for (int i = 0; i < MP; ++i) nv[i] = fmaf(PB[j * MP + i], e[j], nv[i]);
#pragma unroll
            for (int i = 0; i < MP; ++i) e[i] = nv[i];
        };
        // Two traversals of the lane's kIirBK blocks: zero-start end state (lane 0: from the carried state), then -- after the workgroup scan --
        // the replay from the true start state, which writes T_b.  MP <= 8: all block states of a traversal are fetched before the chain starts
        // (a load inside the chain loop pays the L2 round trip per block, and this pass is nothing but latency); MP = 16 keeps the rolled
        // loop (64 more live registers spill).
        auto traverse = [&](bool store) {
            if constexpr (MP <= 8) {
                float zq[kIirBK][MP];
#pragma unroll
                for (int q = 0; q < kIirBK; ++q)
#pragma unroll
                    for (int i = 0; i < MP; ++i) zq[q][i] = b0 + q < nblocks ? zb[(b0 + q) * MP + i] : 0.f;
#pragma unroll
                for (int q = 0; q < kIirBK; ++q) {
                    if (store && b0 + q < nblocks) {
#pragma unroll
                        for (int i = 0; i < MP; ++i) tb[(b0 + q) * MP + i] = e[i];
                    }
                    step(zq[q]);
                }
            } else {
#pragma unroll 1
                for (int q = 0; q < kIirBK; ++q) {
                    if (store && b0 + q < nblocks) {
#pragma unroll
                        for (int i = 0; i < MP; ++i) tb[(b0 + q) * MP + i] = e[i];
                    }
                    float z[MP];
#pragma unroll
                    for (int i = 0; i < MP; ++i) z[i] = b0 + q < nblocks ? zb[(b0 + q) * MP + i] : 0.f;
                    step(z);
                }
            }
        };
#pragma unroll
        for (int i = 0; i < MP; ++i) e[i] = c == 0 ? T[i] : 0.f;
        traverse(false);
#pragma unroll
        for (int i = 0; i < MP; ++i) sv[i * kIirBB + c] = e[i];
        __syncthreads();
        iir_block_scan<MP, kIirBB, kIirBRounds>(sv, PK); // sv[.][c] = state at the END of the lane's last block
#pragma unroll
        for (int i = 0; i < MP; ++i) e[i] = c == 0 ? T[i] : sv[i * kIirBB + c - 1];
        traverse(true);
        __syncthreads();
        if (c < MP) T[c] = sv[c * kIirBB + kIirBB - 1];
        __syncthreads();
    }
}

// pass Y: true start state per chunk, re-run, coalesced store.  The lane owning the last sample stores the carried state.
template <int ORD, int NSEC>
__global__ __launch_bounds__(kIirBS) void iir_pass_y(const float* __restrict__ x, float* __restrict__ y, long n, IirCoef<ORD, NSEC> coef, const float* __restrict__ phi,
                                                      const float* __restrict__ zc, const float* __restrict__ tb, float* __restrict__ state_out) {
    constexpr int    MP = ORD * NSEC;
    __shared__ float tile[kIirBS * (kIirL + 1)];
    __shared__ float sv[MP * kIirBS];
    __shared__ __attribute__((aligned(16))) float pl[kIirRounds * MP * MP];
    const int        c     = threadIdx.x;
    const long       base  = (long)blockIdx.x * kIirBS * kIirL;
    const long       chunk = (long)blockIdx.x * kIirBS + c;
    iir_load_phi<MP, kIirRounds>(pl, phi);
    iir_stage_tile(tile, x, base, n);
    // seed: I_0 = Phi_L * T_b + z_0, I_c = z_c
    const float* Tb = tb + (long)blockIdx.x * MP;
#pragma unroll
    for (int i = 0; i < MP; ++i) {
        float v = zc[chunk * MP + i];
        if (c == 0) {
#pragma unroll
            for (int j = 0; j < MP; ++j) v = fmaf(phi[i * MP + j], Tb[j], v); // round-0 matrix == Phi_L
        }
        sv[i * kIirBS + c] = v;
    }
    __syncthreads();
    iir_block_scan<MP, kIirBS, kIirRounds>(sv, pl);
    float st[NSEC][ORD];
#pragma unroll
    for (int s = 0; s < NSEC; ++s)
#pragma unroll
        for (int j = 0; j < ORD; ++j) st[s][j] = c == 0 ? Tb[s * ORD + j] : sv[(s * ORD + j) * kIirBS + c - 1];
    float*     row  = tile + c * (kIirL + 1);
    const long cbeg = base + (long)c * kIirL;
    const int  len  = (int)(n - cbeg < kIirL ? (n - cbeg < 0 ? 0 : n - cbeg) : kIirL);
    if (len == kIirL) {
#pragma unroll 4
        for (int i = 0; i < kIirL; ++i) row[i] = iir_step<ORD, NSEC>(coef, st, row[i]);
    } else {
        for (int i = 0; i < len; ++i) row[i] = iir_step<ORD, NSEC>(coef, st, row[i]);
    }
    if (len > 0 && cbeg + len == n) { // this lane consumed the last sample of the span
#pragma unroll
        for (int s = 0; s < NSEC; ++s)
#pragma unroll
            for (int j = 0; j < ORD; ++j) state_out[s * ORD + j] = st[s][j];
    }
    __syncthreads();
    iir_unstage_tile(tile, y, base, n);
}

// ------------------------------------------------------------------------------------------------ single pass (MP <= 8)
// The three passes above read x twice, write and re-read the per-chunk states (MP floats per 32 samples) and chain the blocks in a
// one-workgroup kernel.  One pass with a decoupled look-back (the affine-map version of the chained scan) does each of them once:
//   1. stage the tile, zero-state run per chunk, scan over the 64 chunks of every wave IN REGISTERS (no barriers; the four waves are
//      stitched together in step 5)
//   2. publish the block's zero-state end state Z_b (flag 1)
//   3. wave 0 looks back over the predecessors, 64 at a time: lane l reads block b-1-l; aggregates contribute Phi_B^l Z, the first
//      block that already knows its true end state P (flag 2) closes the sum:  T_b = sum_{l<w} Phi_B^l Z_{b-1-l} + Phi_B^w P_{b-1-w}
//      (Phi_B^0..63 from a table; a window without an inclusive block is folded in through Phi_B^64 and the next window follows)
//   4. publish P_b = Phi_B T_b + Z_b (flag 2)
//   5. wave w starts from T^w = Phi_L^{64 w} T_b + (the waves before it); its chunk l = 16 a + r starts from S_{l-1} + Phi_L^{16a} (Phi_L^r T^w):
//      two small tables instead of a second scan
//   6. re-run the chunk from its true start state, coalesced store.
// Where a block's ~19 us go (s_memrealtime stamps, -DGR4_IIR_TIMING + tools/iir_stamps.py, 4 biquads): stage 1.8, run 1.9, scan 3.1,
// LOOK-BACK 7.6, start states 1.4, re-run 2.1, store 0.6.  A poll of the status words is a ~2 us round trip while the streaming
// tiles saturate the memory system, and a block needs 3.8 of them (2.2 windows; the nearest predecessors publish Z within +-2 us of
// this block).  Wider look-backs (2 or 4 waves polling 128 / 256 predecessors per round trip, partial sums handed over through LDS)
// were built and measured SLOWER (212 / 173 vs 255 Gsamples/s): the extra polling traffic lengthens every round trip by more than
// the saved windows are worth.  Cheaper polls did not help either: records {Z, P} of 2 MP consecutive words read cooperatively (a wave-wide
// load covers whole records: 64 memory requests per round instead of 1024, words handed to the owning lane through LDS) and the window sum as a
// six-level tree with Phi_B^(2^k) from LDS (no per-lane Phi_B^l, 106 VGPRs) left the raw round trip at ~2.5 us -- it is latency, not
// request count -- while the LDS transposes and the tree lengthen every window; a longer look-back puts the nearest true end state
// further back (2.3 -> 3.0 windows), which lengthens it again: 256 and 224 instead of 275 Gsamples/s.  Requesting the NEXT window's words together
// with this window's first poll (its round trip would be over when the walk gets there): 235 / 374.  Everything that makes a block poll more
// per round trip has measured slower than the plain 64-lane walk.  Persistent workgroups (tables loaded once, the next tile's ticket and samples prefetched into registers during
// the re-run) were built too: stage 1.8 -> 0.7 us, but no gain in throughput (272 / 429 vs 275 / 423 Gsamples/s, spill-free); the ticket must
// not be drawn before the look-back is over (a ticket held by a block that is still busy makes every successor wait for its Z: 192).
// Block indices are tickets drawn at the start (a block only ever waits for blocks that already run), status words and ticket are zeroed per call.
// Waiting is bounded: a waiter that gives up (never observed) raises a page-locked error word; the span's output is then INVALID and the handle says so:
// gr4hip_iir_status() after the caller's own stream synchronisation, or at the latest the next process / reset call, returns GR4HIP_RUNTIME_ERROR
// (nothing is recomputed behind the caller's back; GR4HIP_IIR_THREE_PASS=1 selects the three-pass kernels, which cannot time out).
struct IirOnePassArgs {
    const float* x;
    float*       y;
    long         n;
    const float* phi;      // [kIirRounds][MP][MP]  Phi_{L 2^k}
    const float* plr;      // [16][MP][MP]          Phi_L^r
    const float* pl16;     // [16][MP][MP]          Phi_L^{16 a}
    const float* pb;       // [65][MP][MP]          Phi_B^l, l = 0 .. 64
    const float* state_in; // [MP]
    float*       state_out;
    // block status: every component is ONE 64-bit word {value, tag = 1}, written by one atomic store.  A separate flag word behind plain
    // or write-through data stores can become visible before the data (measured: wrong states at 8 floats per block); a device-scope
    // release / acquire pair closes that hole but writes back and invalidates the whole L2 of the XCD per block (4x slower than three
    // passes).  Self-validating words need no ordering at all.  Zeroed before every launch.
    unsigned long long* st_z; // [nblocks][MP]  zero-state end states
    unsigned long long* st_p; // [nblocks][MP]  true end states
    unsigned*           ticket; // block tickets
#ifdef GR4_IIR_TIMING
    unsigned long long* dbgc; // [nblocks][16] stamps
#endif
    unsigned*           err;    // page-locked host word (device view): set when a look-back gave up; the host reports it on the handle's next call
};

__device__ __forceinline__ void iir_status_put(unsigned long long* p, float v) {
    __hip_atomic_store(p, ((unsigned long long)1 << 32) | __float_as_uint(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ unsigned long long iir_status_get(const unsigned long long* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// inclusive scan over the 64 chunks of one wave, in registers:  e(lane) <- sum_{i<=lane} Phi_L^{lane-i} e(i).  No LDS state and no
// barriers (the block-wide LDS scan above cost 16 barriers and 4.9 us of a 20 us block in the single-pass kernel); the four waves of a
// block are stitched together afterwards with the same two-table trick that applies T_b.
// A cascade is block lower triangular in its sections (section s sees the states of sections <= s only), and so is every power of Phi:
// the matrix-vector products of the single-pass kernel skip the structural zeros (40 instead of 64 multiply-adds for four biquads).
template <int MP, int ORD>
__device__ __forceinline__ void iir_wave_scan(float (&e)[MP], const float* pl /*LDS [round][MP(j)][MP(i)]: Phi_L^(2^k) transposed*/, int lane) {
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int    off = 1 << k;
        const float* P   = pl + k * MP * MP;
        float        p[MP];
#pragma unroll
        for (int j = 0; j < MP; ++j) p[j] = __shfl_up(e[j], off);
        if (lane >= off) {
#pragma unroll
            for (int j = 0; j < MP; ++j)
#pragma unroll
                for (int i = (j / ORD) * ORD; i < MP; ++i) e[i] = fmaf(P[j * MP + i], p[j], e[i]);
        }
    }
}

// developer instrumentation (-DGR4_IIR_TIMING): lane 0 of every block stamps s_memrealtime (100 MHz) at phase boundaries; the host dumps
// them to /tmp/iir_stamps.txt after every launch (tools/iir_stamps.py reads that)
#ifdef GR4_IIR_TIMING
#define IIR_STAMP(k) do { if (c == 0) { unsigned long long t_; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_)::"memory"); a.dbgc[bid_s * 16 + (k)] = t_; } } while (0)
#define IIR_COUNT(k, v) do { if (c == 0) a.dbgc[bid_s * 16 + (k)] = (v); } while (0)
#else
#define IIR_STAMP(k) do { } while (0)
#define IIR_COUNT(k, v) do { } while (0)
#endif
template <int ORD, int NSEC>
__global__ __launch_bounds__(kIirBS) void iir_onepass_kernel(IirOnePassArgs a, IirCoef<ORD, NSEC> coef) {
    constexpr int    MP = ORD * NSEC;
    static_assert(MP <= 8, "the look-back tables are sized for MP <= 8");
    __shared__ float tile[kIirBS * (kIirL + 1)];
    __shared__ __attribute__((aligned(16))) float pl[kIirRounds * MP * MP];
    __shared__ float p16[16 * MP * MP]; // Phi_L^{16 a}, row-major
    __shared__ float plr[16 * MP * MP]; // Phi_L^r (needed the moment T_b is known: a global read there is a cold round trip on the block's critical path)
    __shared__ float wv[4 * 16 * MP];   // Phi_L^r T^w (T^w: state at the start of wave w's first chunk)
    __shared__ float wz[4 * MP], Tw[4 * MP]; // zero-state end state of every wave's 64 chunks; T^w
    __shared__ float R[MP * MP], R2[MP * MP];
    __shared__ unsigned bid_s;
    const int c = threadIdx.x, lane = c & 63, wave = c >> 6;
    __builtin_amdgcn_s_setprio(3); // until Z_b is out (every successor waits for it) this block goes first on its SIMDs: +1 .. 2 %
    if (c == 0) bid_s = atomicAdd(a.ticket, 1u);
    iir_load_phi<MP, kIirRounds>(pl, a.phi);
    for (int e = c; e < 16 * MP * MP; e += kIirBS) {
        p16[e] = a.pl16[e];
        plr[e] = a.plr[e];
    }
    __syncthreads();
    const long b    = bid_s;
    const long base = b * kIirBS * kIirL;
    IIR_STAMP(0);
    iir_stage_tile(tile, a.x, base, a.n);
    __syncthreads();
    IIR_STAMP(1);
    // ---- 1. zero-state run and in-block scan
    float st[NSEC][ORD];
#pragma unroll
    for (int s = 0; s < NSEC; ++s)
#pragma unroll
        for (int j = 0; j < ORD; ++j) st[s][j] = 0.f;
    float* row = tile + c * (kIirL + 1);
#pragma unroll 4
    for (int i = 0; i < kIirL; ++i) (void)iir_step<ORD, NSEC>(coef, st, row[i]);
    float e[MP], ex[MP]; // zero-state end state of the wave's chunks 0..lane (inclusive / exclusive)
#pragma unroll
    for (int s = 0; s < NSEC; ++s)
#pragma unroll
        for (int j = 0; j < ORD; ++j) e[s * ORD + j] = st[s][j];
    IIR_STAMP(2);
    iir_wave_scan<MP, ORD>(e, pl, lane);
#pragma unroll
    for (int i = 0; i < MP; ++i) {
        ex[i] = __shfl_up(e[i], 1);
        if (lane == 0) ex[i] = 0.f;
        if (lane == 63) wz[wave * MP + i] = e[i];
    }
    __syncthreads();
    IIR_STAMP(3);
    if (wave != 0) __builtin_amdgcn_s_setprio(0);
    // ---- 2..4: wave 0 publishes, looks back, publishes again
    float zb = 0.f; // wave 0, lane i < MP: component i of Z_b
    float p1[MP];   // wave 0: row `lane` of Phi_B (for P_b below; requested here, used after the look-back)
    if (wave == 0) {
        if (lane < MP) {
            zb = wz[3 * MP + lane]; // Z_b = sum_w Phi_L^{64 (3 - w)} wz[w]
#pragma unroll
            for (int w = 0; w < 3; ++w) {
                const float* P = p16 + 4 * (3 - w) * MP * MP;
#pragma unroll
                for (int k = 0; k < MP; ++k) zb = fmaf(P[lane * MP + k], wz[w * MP + k], zb);
            }
            if (b > 0) iir_status_put(a.st_z + b * MP + lane, zb);
        }
        __builtin_amdgcn_s_setprio(0);
#pragma unroll
        for (int k = 0; k < MP; ++k) p1[k] = lane < MP ? a.pb[1L * MP * MP + lane * MP + k] : 0.f;
    }
    float tv = 0.f; // wave 0, lane i < MP: component i of T_b
    if (b == 0) {
        if (wave == 0 && lane < MP) tv = a.state_in[lane];
    } else if (wave == 0) {
        if (lane < MP * MP) R[lane] = (lane / MP == lane % MP) ? 1.f : 0.f; // R = Phi_B^{64 it}: identity for the first window
        float acc = 0.f;   // wave 0, lane i < MP
        float Pl[MP * MP]; // Phi_B^lane: requested before the polling starts, so its L2 round trip overlaps the first poll
        {
            const float* P = a.pb + (long)lane * MP * MP;
#pragma unroll
            for (int e = 0; e < MP * MP; ++e) Pl[e] = P[e];
        }
        // Ticket order is start order only to +-2 us: polled at once, half of the nearest predecessors have not published their Z yet and the
        // round trip is spent on a retry.  A short nap first (0.8 / 1.5 us) is cheaper: +2 % (MP = 8), +8 % (MP = 4).
        if constexpr (MP <= 4) __builtin_amdgcn_s_sleep(50); else __builtin_amdgcn_s_sleep(25);
        int nround = 0, it = 0;
        for (;; ++it) {
            const long j     = b - 1 - 64L * it - lane;
            const bool valid = j >= 0;
            unsigned   flag  = 0; // 1: aggregate, 2: inclusive
            float      z[MP];
#pragma unroll
            for (int i = 0; i < MP; ++i) z[i] = 0.f;
            // Polling rounds are wave-synchronous: all 2 MP status words of block j in ONE round trip (word after word cost 2 + MP - 1
            // dependent latencies), a state is taken only when every word of it validates, and the window is complete as soon as
            // every lane IN FRONT OF the first inclusive state has data -- a late block behind that point is not waited for.
            const unsigned long long* sp = a.st_p + j * MP;
            const unsigned long long* sz = a.st_z + j * MP;
            int                       w  = 64; // first lane with an inclusive state
            for (int spins = 0;; ++spins) {
                if (valid && flag == 0) {
                    unsigned long long wp[MP], wq[MP];
#pragma unroll
                    for (int i = 0; i < MP; ++i) wp[i] = iir_status_get(sp + i);
#pragma unroll
                    for (int i = 0; i < MP; ++i) wq[i] = iir_status_get(sz + i);
                    unsigned long long okp = wp[0], okz = wq[0];
#pragma unroll
                    for (int i = 1; i < MP; ++i) { okp &= wp[i]; okz &= wq[i]; }
                    if (okp >> 32) { // the true end state wins
                        flag = 2;
#pragma unroll
                        for (int i = 0; i < MP; ++i) z[i] = __uint_as_float((unsigned)wp[i]);
                    } else if (okz >> 32) {
                        flag = 1;
#pragma unroll
                        for (int i = 0; i < MP; ++i) z[i] = __uint_as_float((unsigned)wq[i]);
                    }
                }
                if (nround == 0) IIR_STAMP(8);
                ++nround;
                const unsigned long long incl = __ballot(valid && flag == 2);
                w = incl ? __ffsll((long long)incl) - 1 : 64;
                if (!__ballot(valid && flag == 0 && lane < w)) break;
                if (spins > (1 << 22)) { __hip_atomic_store(a.err, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; } // give up instead of hanging (never observed)
                __builtin_amdgcn_s_sleep(2);
            }
            if (it == 0) IIR_STAMP(9);
            float v[MP];
#pragma unroll
            for (int i = 0; i < MP; ++i) v[i] = 0.f;
            if (valid && lane <= w) { // Phi_B^lane * state[j]
#pragma unroll
                for (int i = 0; i < MP; ++i)
#pragma unroll
                    for (int k = 0; k < (i / ORD + 1) * ORD; ++k) v[i] = fmaf(Pl[i * MP + k], z[k], v[i]);
            }
#pragma unroll
            for (int i = 0; i < MP; ++i) { // wave sum (every lane ends with the total)
#pragma unroll
                for (int off = 32; off >= 1; off >>= 1) v[i] += __shfl_xor(v[i], off);
            }
            if (lane < MP) { // acc += R * sum
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < MP; ++k) t = fmaf(R[lane * MP + k], v[k], t);
                acc += t;
            }
            if (w < 64) break;
            // no inclusive state in this iteration: R <- Phi_B^64 * R ... as both are powers of one matrix the order does not matter
            if (lane < MP * MP) {
                const int    i = lane / MP, k = lane % MP;
                const float* P = a.pb + 64L * MP * MP;
                float        t = 0.f;
#pragma unroll
                for (int m = 0; m < MP; ++m) t = fmaf(P[i * MP + m], R[m * MP + k], t);
                R2[lane] = t;
            }
            __builtin_amdgcn_wave_barrier(); // (one wave: LDS accesses complete in program order; this only pins the compiler's order)
            if (lane < MP * MP) R[lane] = R2[lane];
            __builtin_amdgcn_wave_barrier();
        }
        tv = acc;
        IIR_COUNT(10, it + 1);
        IIR_COUNT(11, nround);
    }
    if (wave == 0) {
        IIR_STAMP(4);
        float tvk[MP];
#pragma unroll
        for (int k = 0; k < MP; ++k) tvk[k] = __shfl(tv, k);
        // P_b = Phi_B T_b + Z_b
        if (lane < MP) {
            float pv = zb;
#pragma unroll
            for (int k = 0; k < MP; ++k) pv = fmaf(p1[k], tvk[k], pv);
            iir_status_put(a.st_p + b * MP + lane, pv);
        }
        // T^w = Phi_L^{64 w} T_b + sum_{w' < w} Phi_L^{64 (w - 1 - w')} wz[w']
        if (lane < 4 * MP) {
            const int w = lane / MP, i = lane % MP;
            float     t = 0.f;
            {
                const float* P = p16 + 4 * w * MP * MP;
#pragma unroll
                for (int k = 0; k < MP; ++k) t = fmaf(P[i * MP + k], tvk[k], t);
            }
            for (int wp = 0; wp < w; ++wp) {
                const float* P = p16 + 4 * (w - 1 - wp) * MP * MP;
#pragma unroll
                for (int k = 0; k < MP; ++k) t = fmaf(P[i * MP + k], wz[wp * MP + k], t);
            }
            Tw[lane] = t;
        }
    }
    __syncthreads();
    // ---- 5. wv[w][r] = Phi_L^r T^w (r < 16), then every chunk's start state
    for (int q = c; q < 4 * 16 * MP; q += kIirBS) {
        const int    w = q / (16 * MP), r = (q / MP) % 16, i = q % MP;
        const float* P = plr + r * MP * MP;
        float        t = 0.f;
#pragma unroll
        for (int k = 0; k < MP; ++k) t = fmaf(P[i * MP + k], Tw[w * MP + k], t);
        wv[q] = t;
    }
    __syncthreads();
    {
        const int    aa = lane >> 4, r = lane & 15; // start of the wave's chunk `lane`: exponent 16 aa + r
        const float* P = p16 + aa * MP * MP;
        const float* w = wv + (wave * 16 + r) * MP;
#pragma unroll
        for (int s = 0; s < NSEC; ++s)
#pragma unroll
            for (int j = 0; j < ORD; ++j) {
                const int i = s * ORD + j;
                float     t = ex[i];
#pragma unroll
                for (int k = 0; k < (s + 1) * ORD; ++k) t = fmaf(P[i * MP + k], w[k], t);
                st[s][j] = t;
            }
    }
    IIR_STAMP(5);
    // ---- 6. re-run from the true start state
    const long cbeg = base + (long)c * kIirL;
    const int  len  = (int)(a.n - cbeg < kIirL ? (a.n - cbeg < 0 ? 0 : a.n - cbeg) : kIirL);
    if (len == kIirL) {
#pragma unroll 4
        for (int i = 0; i < kIirL; ++i) row[i] = iir_step<ORD, NSEC>(coef, st, row[i]);
    } else {
        for (int i = 0; i < len; ++i) row[i] = iir_step<ORD, NSEC>(coef, st, row[i]);
    }
    if (len > 0 && cbeg + len == a.n) { // this lane consumed the last sample of the span
#pragma unroll
        for (int s = 0; s < NSEC; ++s)
#pragma unroll
            for (int j = 0; j < ORD; ++j) a.state_out[s * ORD + j] = st[s][j];
    }
    __syncthreads();
    IIR_STAMP(6);
    iir_unstage_tile(tile, a.y, base, a.n);
    IIR_STAMP(7);
}


// ------------------------------------------------------------------------------------------------ segment-sequential kernel (long spans, fading memory)
// The look-back above is ~45 % of a block's residence and every cheaper-poll variant measured slower.  For a long span there is a way around the
// dependency chain altogether: a workgroup takes a CONTIGUOUS run of tiles and walks it in order, the state at a tile's start is simply what the previous
// tile left (kept in LDS) -- no tickets, no status words, no polling, nothing that can time out.  Only the state at the START of a workgroup's run is
// unknown; a stable filter forgets: after W samples an error in the start state has shrunk by ||Phi^W||.  The host picks the smallest warm-up of 1, 2 or 4
// tiles with ||Phi_B^w||_inf <= 1e-8 (below float32 resolution of the state; the parity bar is 1e-5) and every run but the first starts w tiles early from
// the zero state with stores and the second recurrence pass switched off (a warm-up tile costs the zero-state pass and the scan only).  Filters whose memory
// does not fade within 4 tiles = 32768 samples (poles within ~6e-4 of the unit circle) keep the look-back kernel, and so do short spans.
struct IirSeqArgs {
    const float* x;
    float*       y;
    long         n;
    const float* phi;  // [kIirRounds][MP][MP]  Phi_{L 2^k}
    const float* plr;  // [16][MP][MP]          Phi_L^r
    const float* pl16; // [16][MP][MP]          Phi_L^{16 a}
    const float* pb;   // [65][MP][MP]          Phi_B^l
    const float* state_in;
    float*       state_out;
    long         tiles_per_wg;
    int          warm_tiles;
    int          warm_chunks; // warm_tiles == 1: the last this many 32-sample chunks of the warm-up tile are enough (a multiple of 32, <= 256)
    int          nt;          // streaming hints on the tiles' loads and stores (spans beyond the memory-side cache)
};

template <int ORD, int NSEC>
__global__ __launch_bounds__(kIirBS, 4) void iir_seq_kernel(IirSeqArgs a, IirCoef<ORD, NSEC> coef) {
    constexpr int MP = ORD * NSEC;
    static_assert(MP <= 8, "tables are sized for MP <= 8");
    __shared__ float tile[kIirBS * (kIirL + 1)];
    __shared__ __attribute__((aligned(16))) float pl[kIirRounds * MP * MP];
    const float* __restrict__ p16 = a.pl16; // Phi_L^{16 a} and Phi_L^r stay in global memory (L1 / L2 resident, read a few times per tile): the 8 KiB they took
    const float* __restrict__ plr = a.plr;  // in LDS were what kept a fourth workgroup off the CU
    __shared__ float wv[4 * 16 * MP];
    __shared__ float wz[4 * MP], Tw[4 * MP], Tcar[MP];
    const int  c = threadIdx.x, lane = c & 63, wave = c >> 6;
    const long ntiles = (a.n + (long)kIirBS * kIirL - 1) / ((long)kIirBS * kIirL);
    const long t0 = (long)blockIdx.x * a.tiles_per_wg, t1 = t0 + a.tiles_per_wg < ntiles ? t0 + a.tiles_per_wg : ntiles;
    if (t0 >= ntiles) return;
    iir_load_phi<MP, kIirRounds>(pl, a.phi);
    long tb = t0 - a.warm_tiles;
    if (c < MP) Tcar[c] = tb <= 0 ? a.state_in[c] : 0.f; // the first run starts from the handle's carried state, exactly
    if (tb < 0) tb = 0;
    float p1[MP]; // wave 0, lane < MP: row `lane` of Phi_B
#pragma unroll
    for (int k = 0; k < MP; ++k) p1[k] = (wave == 0 && lane < MP) ? a.pb[1L * MP * MP + lane * MP + k] : 0.f;
    __syncthreads();
    for (long b = tb; b < t1; ++b) {
        const bool emit = b >= t0; // warm-up tiles only advance the state
        const long base = b * kIirBS * kIirL;
        // (round 5) a warm-up tile exists to leave the state at its END, and the filter has forgotten everything older than warm_chunks chunks by then (||Phi_L^wc|| <= 1e-8):
        // only the tile's last wc chunks are read from HBM and run -- the lanes in front of them keep the zero state (for Butterworth-8 at fc = 0.05 that is 32 chunks of
        // 256: an eighth of the tile's traffic and of its zero-state instructions; a run of two tiles per workgroup paid 3 reads and 3 zero-state passes for 2 tiles)
        const int  first = emit ? 0 : kIirBS - a.warm_chunks; // first chunk (lane) of the tile that takes part
        if (first == 0) iir_stage_tile(tile, a.x, base, a.n, a.nt != 0);
        else if (base + (long)kIirBS * kIirL <= a.n && (reinterpret_cast<uintptr_t>(a.x + base) & 15) == 0) { // (first is a multiple of 32 chunks = 4 of the lane's 8 float4 slots... slot q covers chunks 32 q .. 32 q + 31)
#pragma unroll
            for (int q = 0; q < kIirL / 4; ++q) {
                if (32 * q < first) continue; // (uniform)
                const float4 v = reinterpret_cast<const float4*>(a.x + base)[q * kIirBS + threadIdx.x];
                const int    s_ = 4 * (q * kIirBS + threadIdx.x);
                float*       d = tile + (s_ / kIirL) * (kIirL + 1) + (s_ % kIirL);
                d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
            }
        } else {
            for (int s_ = first * kIirL + threadIdx.x; s_ < kIirBS * kIirL; s_ += kIirBS) {
                const long i = base + s_;
                tile[(s_ / kIirL) * (kIirL + 1) + (s_ % kIirL)] = i < a.n ? a.x[i] : 0.f;
            }
        }
        __syncthreads();
        // ---- 1. zero-state run and in-wave scan
        float st[NSEC][ORD];
#pragma unroll
        for (int s = 0; s < NSEC; ++s)
#pragma unroll
            for (int j = 0; j < ORD; ++j) st[s][j] = 0.f;
        float* row = tile + c * (kIirL + 1);
        if (c >= first) {
#pragma unroll 4
            for (int i = 0; i < kIirL; ++i) (void)iir_step<ORD, NSEC>(coef, st, row[i]);
        }
        float e[MP], ex[MP];
#pragma unroll
        for (int s = 0; s < NSEC; ++s)
#pragma unroll
            for (int j = 0; j < ORD; ++j) e[s * ORD + j] = st[s][j];
        iir_wave_scan<MP, ORD>(e, pl, lane);
#pragma unroll
        for (int i = 0; i < MP; ++i) {
            ex[i] = __shfl_up(e[i], 1);
            if (lane == 0) ex[i] = 0.f;
            if (lane == 63) wz[wave * MP + i] = e[i];
        }
        __syncthreads();
        // ---- 2. wave 0: Z_b, the next tile's start state P_b = Phi_B T_b + Z_b, and the four waves' start states T^w
        if (wave == 0) {
            float tvk[MP];
#pragma unroll
            for (int k = 0; k < MP; ++k) tvk[k] = Tcar[k];
            if (lane < MP) {
                float zb = wz[3 * MP + lane]; // Z_b = sum_w Phi_L^{64 (3 - w)} wz[w]
#pragma unroll
                for (int w = 0; w < 3; ++w) {
                    const float* P = p16 + 4 * (3 - w) * MP * MP;
#pragma unroll
                    for (int k = 0; k < MP; ++k) zb = fmaf(P[lane * MP + k], wz[w * MP + k], zb);
                }
                float pv = zb;
#pragma unroll
                for (int k = 0; k < MP; ++k) pv = fmaf(p1[k], tvk[k], pv);
                Tcar[lane] = pv; // (every lane of the wave has read Tcar above: one wave, LDS in program order)
            }
            if (emit && lane < 4 * MP) { // T^w = Phi_L^{64 w} T_b + sum_{w' < w} Phi_L^{64 (w - 1 - w')} wz[w']
                const int w = lane / MP, i = lane % MP;
                float     t = 0.f;
                {
                    const float* P = p16 + 4 * w * MP * MP;
#pragma unroll
                    for (int k = 0; k < MP; ++k) t = fmaf(P[i * MP + k], tvk[k], t);
                }
                for (int wp = 0; wp < w; ++wp) {
                    const float* P = p16 + 4 * (w - 1 - wp) * MP * MP;
#pragma unroll
                    for (int k = 0; k < MP; ++k) t = fmaf(P[i * MP + k], wz[wp * MP + k], t);
                }
                Tw[lane] = t;
            }
        }
        __syncthreads();
        if (!emit) continue; // (uniform)
        // ---- 3. wv[w][r] = Phi_L^r T^w (r < 16), then every chunk's start state
        for (int q = c; q < 4 * 16 * MP; q += kIirBS) {
            const int    w = q / (16 * MP), r = (q / MP) % 16, i = q % MP;
            const float* P = plr + r * MP * MP;
            float        t = 0.f;
#pragma unroll
            for (int k = 0; k < MP; ++k) t = fmaf(P[i * MP + k], Tw[w * MP + k], t);
            wv[q] = t;
        }
        __syncthreads();
        {
            const int    aa = lane >> 4, r = lane & 15;
            const float* P = p16 + aa * MP * MP;
            const float* w = wv + (wave * 16 + r) * MP;
#pragma unroll
            for (int s = 0; s < NSEC; ++s)
#pragma unroll
                for (int j = 0; j < ORD; ++j) {
                    const int i = s * ORD + j;
                    float     t = ex[i];
#pragma unroll
                    for (int k = 0; k < (s + 1) * ORD; ++k) t = fmaf(P[i * MP + k], w[k], t);
                    st[s][j] = t;
                }
        }
        // ---- 4. re-run from the true start state.  (Tried instead: keep the zero-state outputs of pass 1 and add the homogeneous response y[k] += sum_j C[k][j] s0[j],
        //      MP multiply-adds per sample with C from scalar loads: 13 % SLOWER at 4 biquads, 3 % at one pole -- the 32 extra LDS writes of pass 1 and the
        //      scalar-load waits in the loop cost more than the second cascade run.)
        const long cbeg = base + (long)c * kIirL;
        const int  len  = (int)(a.n - cbeg < kIirL ? (a.n - cbeg < 0 ? 0 : a.n - cbeg) : kIirL);
        if (len == kIirL) {
#pragma unroll 4
            for (int i = 0; i < kIirL; ++i) row[i] = iir_step<ORD, NSEC>(coef, st, row[i]);
        } else {
            for (int i = 0; i < len; ++i) row[i] = iir_step<ORD, NSEC>(coef, st, row[i]);
        }
        if (len > 0 && cbeg + len == a.n) { // this lane consumed the last sample of the span
#pragma unroll
            for (int s = 0; s < NSEC; ++s)
#pragma unroll
                for (int j = 0; j < ORD; ++j) a.state_out[s * ORD + j] = st[s][j];
        }
        __syncthreads();
        iir_unstage_tile(tile, a.y, base, a.n, a.nt != 0);
        __syncthreads(); // the tile is staged again at the top
    }
}

// ------------------------------------------------------------------------------------------------ sequential float32 kernel (GR4HIP_IIR_SEQUENTIAL_F32)
// The reference's own arithmetic, form by form (detail::computeFilter, FilterTool.hpp:116-158): one lane walks the cascade sample by sample with the section's
// input / output histories in registers, float32 operations in source order (no fused multiply-adds) -- slow (one dependent chain), and exactly as close to
// float64 as the block on the host.  For cascades whose state the parallel-in-time evaluation cannot carry in float32 (ill-conditioned narrow-band designs of
// high order: profiles/r03_fuzz_summary.txt) and for callers who chose a form for its noise behaviour.
constexpr int kIirSeqMaxSec = 8, kIirSeqMaxOrd = 4, kIirSeqChunk = 4096;
struct IirSeqF32Coef {
    float b[kIirSeqMaxSec][kIirSeqMaxOrd + 1], a[kIirSeqMaxSec][kIirSeqMaxOrd + 1]; // zero padded
    int   nsec, nb, na, form;                                                        // nb / na: coefficient counts per section (na == 1: no feedback)
};
// one section, one sample; ih / oh: input / output history, newest first (what push_front + cbegin() give upstream)
template <typename T>
__host__ __device__ inline T iir_form_step(int form, const float* b, const float* a, int nb, int na, T (&ih)[kIirSeqMaxOrd + 1], T (&oh)[kIirSeqMaxOrd + 1], T x) {
#pragma clang fp contract(off)
    const auto push = [](T (&h)[kIirSeqMaxOrd + 1], T v) {
#pragma unroll
        for (int j = kIirSeqMaxOrd; j > 0; --j) h[j] = h[j - 1];
        h[0] = v;
    };
    const auto dot = [](const float* c, int first, int count, const T (&h)[kIirSeqMaxOrd + 1]) { // sum_{k = first}^{count - 1} c[k] h[k - first]
        T acc = T(0);
#pragma unroll
        for (int k = 0; k <= kIirSeqMaxOrd; ++k)
            if (k >= first && k < count) acc = acc + (T)c[k] * h[k - first];
        return acc;
    };
    if (form == GR4HIP_DF_I) {
        push(ih, x);
        const T o = dot(b, 0, nb, ih) - dot(a, 1, na, oh);
        push(oh, o);
        return o;
    }
    if (form == GR4HIP_DF_II) {
        if (na > 1) {
            const T w = x - dot(a, 1, na, ih);
            push(ih, w);
        } else {
            push(ih, x);
        }
        return dot(b, 0, nb, ih);
    }
    if (form == GR4HIP_DF_I_TRANSPOSED) {
        const T v0 = x - dot(a, 1, na, oh);
        push(oh, v0);
        return dot(b, 0, nb, oh);
    }
    const T o = (T)b[0] * x + dot(b, 1, nb, ih) - dot(a, 1, na, oh); // DF_II_TRANSPOSED as the reference writes it
    push(ih, x);
    push(oh, o);
    return o;
}

__global__ __launch_bounds__(64) void iir_sequential_kernel(const float* __restrict__ x, float* __restrict__ y, long n, IirSeqF32Coef c, float* __restrict__ state /*[nsec][2][kIirSeqMaxOrd + 1]*/) {
    __shared__ float buf[kIirSeqChunk];
    const int lane = threadIdx.x;
    float     ih[kIirSeqMaxSec][kIirSeqMaxOrd + 1], oh[kIirSeqMaxSec][kIirSeqMaxOrd + 1];
#pragma unroll
    for (int s = 0; s < kIirSeqMaxSec; ++s)
#pragma unroll
        for (int j = 0; j <= kIirSeqMaxOrd; ++j) {
            ih[s][j] = (lane == 0 && s < c.nsec) ? state[(s * 2 + 0) * (kIirSeqMaxOrd + 1) + j] : 0.f;
            oh[s][j] = (lane == 0 && s < c.nsec) ? state[(s * 2 + 1) * (kIirSeqMaxOrd + 1) + j] : 0.f;
        }
    for (long base = 0; base < n; base += kIirSeqChunk) {
        const int cnt = (int)(n - base < kIirSeqChunk ? n - base : kIirSeqChunk);
        for (int i = lane; i < cnt; i += 64) buf[i] = x[base + i];
        __syncthreads();
        if (lane == 0) {
            for (int i = 0; i < cnt; ++i) {
                float v = buf[i];
#pragma unroll
                for (int s = 0; s < kIirSeqMaxSec; ++s)
                    if (s < c.nsec) v = iir_form_step<float>(c.form, c.b[s], c.a[s], c.nb, c.na, ih[s], oh[s], v);
                buf[i] = v;
            }
        }
        __syncthreads();
        for (int i = lane; i < cnt; i += 64) y[base + i] = buf[i];
        __syncthreads();
    }
    if (lane == 0) {
#pragma unroll
        for (int s = 0; s < kIirSeqMaxSec; ++s)
#pragma unroll
            for (int j = 0; j <= kIirSeqMaxOrd; ++j)
                if (s < c.nsec) {
                    state[(s * 2 + 0) * (kIirSeqMaxOrd + 1) + j] = ih[s][j];
                    state[(s * 2 + 1) * (kIirSeqMaxOrd + 1) + j] = oh[s][j];
                }
    }
}

} // namespace gr4

using namespace gr4;

struct gr4hip_iir {
    int                 form = GR4HIP_DF_II;
    int                 nsec = 0, ord = 2, M = 0; // nsec / M include the identity padding sections
    std::vector<double> b, a; // [nsec][ord+1]
    DeviceBuffer        d_phi;   // [kIirRounds + kIirBShift + kIirBRounds][M][M]: Phi_{L 2^k}
    DeviceBuffer        d_state[2];
    int                 cur = 0;
    DeviceBuffer        d_zc, d_zb, d_tb;
    DeviceBuffer        d_tab;            // one-pass tables: Phi_L^r [16], Phi_L^{16a} [16], Phi_B^l [65]  (M <= 8)
    DeviceBuffer        d_stz;            // one-pass block status words: [nblocks][M] x 2 (+ ticket)
    int                 warm_chunks = 256; // ... and, when one tile is enough, how many of its last 32-sample chunks are (a multiple of 32)
    int                 warm_tiles = 0;   // segment-sequential kernel: warm-up tiles that make a run's unknown start state irrelevant (0: memory does not fade fast enough)
    unsigned*           h_err = nullptr;  // page-locked, device-visible: a look-back that timed out (never observed) is reported by the next call, loudly
    // more than 8 state values (5 .. 8 biquads, 3 .. 4 sections of order 4): two cascades of <= 8 state values run one behind the other through a scratch
    // stream -- the section order and the float32 rounding between sections are the cascade's own, and each half takes the fast kernels (measured: 8 biquads
    // as one 16-state scan 69 Gsamples/s, as 4 + 4 biquads 230)
    gr4hip_iir*         part[2] = {nullptr, nullptr};
    DeviceBuffer        d_mid;
    // the cascade as given (top-level handle): what GR4HIP_IIR_SEQUENTIAL_F32 walks, and what the create-time self-test of GR4HIP_IIR_AUTO compares against
    IirSeqF32Coef       seq{};
    DeviceBuffer        d_seq_state;
    DeviceBuffer        d_fuse_tab;       // the cascade as the decimator's fused launch reads it (fir_decim_fd.hip: 32 floats per section), built on first use
    int                 fuse_warm = -1;   // blocks of 896 samples after which the cascade has forgotten its start state (||Phi_896^w|| <= 1e-8); 0: it does not within 4
    int                 algo = GR4HIP_IIR_AUTO, algo_in_use = GR4HIP_IIR_PARALLEL;
    float               selftest_parallel = -1.f, selftest_f32 = -1.f; // create-time errors (max |.| / output rms against float64) of the parallel kernels and of the sequential float32 form
    bool                top = true;
    // the stream rule (common.hpp): create / reset / set_algo only note that the state is to be zeroed; iir_state_on() enqueues it on the stream of the next call
    bool                zero_state = true, zero_seq = true;
    ~gr4hip_iir() {
        if (h_err) hip_quiet(hipHostFree(h_err));
        delete part[0];
        delete part[1];
    }
};

// host double-precision cascade step (same recurrence) used to build the propagation matrices
static void host_step(const gr4hip_iir* f, std::vector<double>& st, double x) {
    const int O = f->ord;
    for (int s = 0; s < f->nsec; ++s) {
        double w = x;
        for (int j = 0; j < O; ++j) w -= f->a[s * (O + 1) + j + 1] * st[s * O + j];
        double y = f->b[s * (O + 1)] * w;
        for (int j = 0; j < O; ++j) y += f->b[s * (O + 1) + j + 1] * st[s * O + j];
        for (int j = O - 1; j > 0; --j) st[s * O + j] = st[s * O + j - 1];
        st[s * O] = w;
        x         = y;
    }
}

// Phi_steps in double: column j = state after `steps` zero-input steps from the unit state e_j
static std::vector<double> host_phi(const gr4hip_iir* f, long steps) {
    const int           M = f->M;
    std::vector<double> out((size_t)M * M);
    for (int j = 0; j < M; ++j) {
        std::vector<double> st(M, 0.0);
        st[j] = 1.0;
        for (long t = 0; t < steps; ++t) host_step(f, st, 0.0);
        for (int i = 0; i < M; ++i) out[i * M + j] = st[i];
    }
    return out;
}
static std::vector<double> mat_square(const std::vector<double>& A, int M) {
    std::vector<double> C((size_t)M * M, 0.0);
    for (int i = 0; i < M; ++i)
        for (int k = 0; k < M; ++k)
            for (int j = 0; j < M; ++j) C[i * M + j] += A[i * M + k] * A[k * M + j];
    return C;
}

static int iir_take_error(gr4hip_iir* f, const char* where);

template <int ORD, int NSEC>
static int iir_run(gr4hip_iir* f, const float* x, float* y, long n, hipStream_t st) {
    constexpr int MP      = ORD * NSEC;
    const long    nblocks = ceil_div(n, (long)kIirBS * kIirL);
    if constexpr (MP <= 8) {
        // long span + fading memory: contiguous runs of tiles per workgroup, state carried from tile to tile, warm-up instead of look-back
        // (from 16 tiles on the sequential runs beat the look-back at every span length measured: 2^17 .. 2^27 samples, profiles/r02_iir_rates.txt)
        if (f->warm_tiles > 0 && nblocks >= 1 && !dev_switch(kDevIirThreePass) && !dev_switch(kDevIirLookback)) {
            static PerDevice per_device;
            bool             first = false;
            int              dev = -1, n_cu = per_device.current(&first, &dev);
            GR4_REQUIRE(n_cu != 0, "iir: cannot query the current device");
            if (first) { n_cu = -n_cu; per_device.done(dev, n_cu); }
            const long slots = 4L * n_cu; // four resident workgroups per CU (LDS 38.7 KB, <= 128 VGPRs)
            // runs as short as the warm-up itself when the span has fewer tiles than the chip has workgroup slots: half of such a run is warm-up, but every slot
            // works (measured, 4 biquads: 2^22 / 2^23 / 2^24 samples 207 / 278 / 358 Gsamples/s against 67 / 130 / 248 with runs of >= 8 warm-ups)
            long       per   = std::max<long>(f->warm_tiles, ceil_div(nblocks, slots));
            IirCoef<ORD, NSEC> cf{};
            for (int s = 0; s < NSEC; ++s)
                for (int j = 0; j <= ORD; ++j) {
                    cf.b[s][j] = (float)f->b[s * (ORD + 1) + j];
                    cf.a[s][j] = (float)f->a[s * (ORD + 1) + j];
                }
            const float* tab = static_cast<const float*>(f->d_tab.ptr);
            IirSeqArgs   a{};
            a.x = x; a.y = y; a.n = n;
            a.phi = static_cast<const float*>(f->d_phi.ptr);
            a.plr = tab; a.pl16 = tab + 16 * MP * MP; a.pb = tab + 32 * MP * MP;
            a.state_in  = static_cast<const float*>(f->d_state[f->cur].ptr);
            a.state_out = static_cast<float*>(f->d_state[f->cur ^ 1].ptr);
            a.tiles_per_wg = per;
            a.warm_tiles   = f->warm_tiles;
            a.warm_chunks  = f->warm_tiles == 1 ? f->warm_chunks : kIirBS;
            a.nt           = (size_t)n * 2 * sizeof(float) > ((size_t)192 << 20); // input + output beyond what the 256 MB memory-side cache would keep
            hipLaunchKernelGGL((iir_seq_kernel<ORD, NSEC>), dim3((unsigned)ceil_div(nblocks, per)), dim3(kIirBS), 0, st, a, cf);
            GR4_LAUNCH_CHECK();
            f->cur ^= 1;
            return GR4HIP_OK;
        }
        if (!dev_switch(kDevIirThreePass)) { // (developer switch: the three-pass kernels below stay the path for MP = 16)
            if (const int e = iir_take_error(f, "iir_process")) return e; // a previous launch of this handle gave up waiting for a predecessor block
            if (!f->h_err) {
                GR4_HIP_TRY(hipHostMalloc(reinterpret_cast<void**>(&f->h_err), sizeof(unsigned), hipHostMallocMapped));
                *f->h_err = 0;
            }
            unsigned* d_err = nullptr;
            GR4_HIP_TRY(hipHostGetDevicePointer(reinterpret_cast<void**>(&d_err), f->h_err, 0));
            const size_t words = (size_t)nblocks * MP;
#ifdef GR4_IIR_TIMING
            constexpr size_t kTimingWords = 16;
#else
            constexpr size_t kTimingWords = 0;
#endif
            int          rc1   = f->d_stz.ensure((2 * words + 1 + kTimingWords * nblocks) * sizeof(unsigned long long)); // Z words, P words, {ticket, error}
            if (rc1) return rc1;
            GR4_HIP_TRY(hipMemsetAsync(f->d_stz.ptr, 0, (2 * words + 1) * sizeof(unsigned long long), st));
            IirCoef<ORD, NSEC> cf{};
            for (int s = 0; s < NSEC; ++s)
                for (int j = 0; j <= ORD; ++j) {
                    cf.b[s][j] = (float)f->b[s * (ORD + 1) + j];
                    cf.a[s][j] = (float)f->a[s * (ORD + 1) + j];
                }
            const float*   tab = static_cast<const float*>(f->d_tab.ptr);
            IirOnePassArgs a{};
            a.x = x; a.y = y; a.n = n;
            a.phi  = static_cast<const float*>(f->d_phi.ptr);
            a.plr  = tab;
            a.pl16 = tab + 16 * MP * MP;
            a.pb   = tab + 32 * MP * MP;
            a.state_in  = static_cast<const float*>(f->d_state[f->cur].ptr);
            a.state_out = static_cast<float*>(f->d_state[f->cur ^ 1].ptr);
            a.st_z   = static_cast<unsigned long long*>(f->d_stz.ptr);
            a.st_p   = a.st_z + words;
            a.ticket = reinterpret_cast<unsigned*>(a.st_p + words);
            a.err    = d_err;
#ifdef GR4_IIR_TIMING
            a.dbgc = a.st_p + words + 1;
#endif
            hipLaunchKernelGGL((iir_onepass_kernel<ORD, NSEC>), dim3((unsigned)nblocks), dim3(kIirBS), 0, st, a, cf);
            GR4_LAUNCH_CHECK();
#ifdef GR4_IIR_TIMING
            {
                std::vector<unsigned long long> h(16 * nblocks);
                hip_quiet(hipStreamSynchronize(st));
                (void)hipMemcpy(h.data(), a.dbgc, h.size() * 8, hipMemcpyDeviceToHost);
                FILE* fp = fopen("/tmp/iir_stamps.txt", "w");
                for (long i = 0; i < nblocks; ++i) { for (int k = 0; k < 16; ++k) fprintf(fp, "%llu ", h[i * 16 + k]); fprintf(fp, "\n"); }
                fclose(fp);
            }
#endif
            f->cur ^= 1;
            return GR4HIP_OK;
        }
    }
    int           rc      = f->d_zc.ensure((size_t)nblocks * kIirBS * MP * sizeof(float));
    if (!rc) rc = f->d_zb.ensure((size_t)nblocks * MP * sizeof(float));
    if (!rc) rc = f->d_tb.ensure((size_t)nblocks * MP * sizeof(float));
    if (rc) return rc;
    IirCoef<ORD, NSEC> coef{};
    for (int s = 0; s < NSEC; ++s)
        for (int j = 0; j <= ORD; ++j) {
            coef.b[s][j] = (float)f->b[s * (ORD + 1) + j];
            coef.a[s][j] = (float)f->a[s * (ORD + 1) + j];
        }
    const float* phi  = static_cast<const float*>(f->d_phi.ptr);
    const float* phiB = phi + (size_t)kIirRounds * MP * MP; // Phi_B^(2^k), k = 0 .. kIirBRounds-1
    hipLaunchKernelGGL((iir_pass_z<ORD, NSEC>), dim3((unsigned)nblocks), dim3(kIirBS), 0, st, x, n, coef, phi, (float*)f->d_zc.ptr, (float*)f->d_zb.ptr);
    GR4_LAUNCH_CHECK();
    hipLaunchKernelGGL(iir_pass_b<MP>, dim3(1), dim3(kIirBB), 0, st, (const float*)f->d_state[f->cur].ptr, phiB, (const float*)f->d_zb.ptr, (float*)f->d_tb.ptr, nblocks);
    GR4_LAUNCH_CHECK();
    hipLaunchKernelGGL((iir_pass_y<ORD, NSEC>), dim3((unsigned)nblocks), dim3(kIirBS), 0, st, x, y, n, coef, phi, (const float*)f->d_zc.ptr, (const float*)f->d_tb.ptr,
                       (float*)f->d_state[f->cur ^ 1].ptr);
    GR4_LAUNCH_CHECK();
    f->cur ^= 1;
    return GR4HIP_OK;
}

extern "C" {

static int iir_create_impl(gr4hip_iir_t** out, int form, size_t nsections, const float* h_b, size_t nb, const float* h_a, size_t na, bool top);
static int iir_selftest(gr4hip_iir* f);
static int iir_state_on(gr4hip_iir* f, hipStream_t st);
struct IirVerdict { int algo; float e_par, e_f32; };
static std::map<std::vector<uint32_t>, IirVerdict> g_selftest; // keyed on bit patterns (a strict weak order whatever the values) + the device; bounded (kSelftestCacheMax)
constexpr size_t                                  kSelftestCacheMax = 4096; // (the errors are ratios to the output rms: a gain on the numerators leaves them as they are)
static std::mutex                               g_selftest_mu;
static int iir_process_parallel(gr4hip_iir_t* f, const float* d_in, size_t n, float* d_out, gr4hip_stream_t stream);

int gr4hip_iir_create(gr4hip_iir_t** out, int form, size_t nsections, const float* h_b, size_t nb, const float* h_a, size_t na) {
    return iir_create_impl(out, form, nsections, h_b, nb, h_a, na, true);
}

// the top-level handle keeps the cascade as given (for the sequential float32 kernel and the create-time self-test)
static int iir_finish_top(gr4hip_iir* f, int form, size_t nsections, const float* h_b, size_t nb, const float* h_a, size_t na) {
    f->top      = true;
    f->seq      = IirSeqF32Coef{};
    f->seq.nsec = (int)nsections;
    f->seq.nb   = (int)nb;
    f->seq.na   = (int)na;
    f->seq.form = form;
    for (size_t s = 0; s < nsections; ++s) {
        for (size_t j = 0; j < nb; ++j) f->seq.b[s][j] = h_b[s * nb + j];
        for (size_t j = 0; j < na; ++j) f->seq.a[s][j] = h_a[s * na + j];
    }
    int rc = f->d_seq_state.ensure((size_t)kIirSeqMaxSec * 2 * (kIirSeqMaxOrd + 1) * sizeof(float));
    if (rc) return rc;
    f->zero_seq = true;
    return iir_selftest(f);
}

static int iir_create_impl(gr4hip_iir_t** out, int form, size_t nsections, const float* h_b, size_t nb, const float* h_a, size_t na, bool top) {
    GR4_REQUIRE(out, "iir: null output handle");
    GR4_REQUIRE(form >= GR4HIP_DF_I && form <= GR4HIP_DF_II_TRANSPOSED, "iir: unknown form %d", form);
    GR4_REQUIRE(nsections >= 1 && h_b && h_a && nb >= 1 && na >= 1, "iir: need >= 1 section and non-empty b, a");
    const size_t order = std::max(nb, na) - 1;
    const int    ord   = order <= 2 ? 2 : 4;
    if (order > 4 || nsections * ord > (size_t)kIirMaxM) {
        set_error("iir: %zu sections of order %zu exceed the device path (order <= 4, sections*order <= %d)", nsections, order, kIirMaxM);
        return GR4HIP_UNSUPPORTED;
    }
    auto* f = new (std::nothrow) gr4hip_iir();
    GR4_REQUIRE(f, "out of host memory");
    f->form = form;
    f->ord  = ord;
    if (nsections * ord > 8 && !dev_switch(kDevIirNoSplit)) { // two cascades of <= 8 state values (GR4HIP_IIR_NO_SPLIT: the 16-state kernels, which the tests compare)
        const size_t k = 8 / ord;
        int rc = iir_create_impl(&f->part[0], form, k, h_b, nb, h_a, na, false);
        if (!rc) rc = iir_create_impl(&f->part[1], form, nsections - k, h_b + k * nb, nb, h_a + k * na, na, false);
        f->top = top;
        if (!rc && top) rc = iir_finish_top(f, form, nsections, h_b, nb, h_a, na);
        if (rc) { delete f; return rc; }
        *out = f;
        return GR4HIP_OK;
    }
    const int mp = (int)nsections * ord <= 4 ? 4 : (int)nsections * ord <= 8 ? 8 : 16; // instantiated state counts
    f->nsec = mp / ord;
    f->M    = mp;
    f->b.assign((size_t)f->nsec * (ord + 1), 0.0);
    f->a.assign((size_t)f->nsec * (ord + 1), 0.0);
    for (int s = (int)nsections; s < f->nsec; ++s) f->b[s * (ord + 1)] = 1.0, f->a[s * (ord + 1)] = 1.0; // identity padding sections
    for (int s = 0; s < (int)nsections; ++s) {
        for (size_t j = 0; j < nb; ++j) f->b[s * (ord + 1) + j] = h_b[s * nb + j];
        for (size_t j = 0; j < na; ++j) f->a[s * (ord + 1) + j] = h_a[s * na + j];
        f->a[s * (ord + 1)] = 1.0; // a[0] is assumed 1 (time_domain_filter.hpp:95,102)
    }
    const size_t       mm = (size_t)f->M * f->M;
    // Phi_{L 2^k}: k < kIirRounds are the in-block rounds, k = kIirRounds is Phi_B (B = L * kIirBS samples), the following ones its
    // powers for the block-level scan; Phi_L by stepping the recurrence in double, the rest by repeated squaring
    std::vector<float>  phi((kIirRounds + kIirBShift + kIirBRounds) * mm);
    std::vector<double> Pk = host_phi(f, kIirL);
    for (int k = 0; k < kIirRounds + kIirBShift + kIirBRounds; ++k) {
        for (size_t i = 0; i < mm; ++i) phi[k * mm + i] = (float)Pk[i];
        Pk = mat_square(Pk, f->M);
    }
    int rc = f->d_phi.ensure(phi.size() * sizeof(float));
    if (!rc) { hipError_t e = upload_fresh(f->d_phi.ptr, phi.data(), phi.size() * sizeof(float)); if (e != hipSuccess) { set_error("iir: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
    for (int k = 0; k < 2 && !rc; ++k) rc = f->d_state[k].ensure(kIirMaxM * sizeof(float));
    if (!rc && f->M <= 8) { // tables of the single-pass kernel, all powers in double
        const int M = f->M;
        auto mul = [M](const std::vector<double>& A, const std::vector<double>& B) {
            std::vector<double> C((size_t)M * M, 0.0);
            for (int i = 0; i < M; ++i)
                for (int k = 0; k < M; ++k)
                    for (int j = 0; j < M; ++j) C[i * M + j] += A[i * M + k] * B[k * M + j];
            return C;
        };
        std::vector<double> I((size_t)M * M, 0.0);
        for (int i = 0; i < M; ++i) I[i * M + i] = 1.0;
        const std::vector<double> PL = host_phi(f, kIirL);
        std::vector<double>       PL16 = PL;
        for (int k = 0; k < 4; ++k) PL16 = mul(PL16, PL16); // Phi_L^16
        std::vector<double> PB = PL16;
        for (int k = 0; k < 4; ++k) PB = mul(PB, PB);       // Phi_L^256 = Phi_B  (kIirBS = 256 chunks per block)
        static_assert(kIirBS == 256, "Phi_B = Phi_L^256");
        std::vector<float> tab((16 + 16 + 65) * mm);
        std::vector<double> cur = I;
        for (int r = 0; r < 16; ++r) { for (size_t i = 0; i < mm; ++i) tab[r * mm + i] = (float)cur[i]; cur = mul(cur, PL); }
        cur = I;
        for (int q = 0; q < 16; ++q) { for (size_t i = 0; i < mm; ++i) tab[(16 + q) * mm + i] = (float)cur[i]; cur = mul(cur, PL16); }
        cur = I;
        for (int l = 0; l < 65; ++l) { for (size_t i = 0; i < mm; ++i) tab[(32 + l) * mm + i] = (float)cur[i]; cur = mul(cur, PB); }
        { // warm-up length of the segment-sequential kernel: smallest w in {1, 2, 4} tiles with ||Phi_B^w||_inf <= 1e-8
            std::vector<double> Pw = PB;
            f->warm_tiles = 0;
            for (int w = 1; w <= 4; w *= 2) {
                double nrm = 0;
                for (int i = 0; i < M; ++i) { double r = 0; for (int j = 0; j < M; ++j) r += std::fabs(Pw[i * M + j]); nrm = std::max(nrm, r); }
                if (nrm <= 1e-8) { f->warm_tiles = w; break; }
                Pw = mul(Pw, Pw);
            }
            f->warm_chunks = kIirBS;
            if (f->warm_tiles == 1) { // one tile is enough: are its last 32 / 64 / 128 chunks?
                std::vector<double> Pc = mul(PL16, PL16); // Phi_L^32
                for (int wc = 32; wc < kIirBS; wc *= 2) {
                    double nrm = 0;
                    for (int i = 0; i < M; ++i) { double r = 0; for (int j = 0; j < M; ++j) r += std::fabs(Pc[i * M + j]); nrm = std::max(nrm, r); }
                    if (nrm <= 1e-8) { f->warm_chunks = wc; break; }
                    Pc = mul(Pc, Pc);
                }
            }
        }
        rc = f->d_tab.ensure(tab.size() * sizeof(float));
        if (!rc) { hipError_t e = upload_fresh(f->d_tab.ptr, tab.data(), tab.size() * sizeof(float)); if (e != hipSuccess) { set_error("iir: upload failed: %s", hipGetErrorString(e)); rc = GR4HIP_RUNTIME_ERROR; } }
    }
    if (rc) { delete f; return rc; }
    f->top = false; // (reset below must not touch the sequential state before it exists)
    rc = gr4hip_iir_reset(f);
    if (!rc && top) rc = iir_finish_top(f, form, nsections, h_b, nb, h_a, na);
    if (rc) { delete f; return rc; }
    f->top = top;
    *out = f;
    return GR4HIP_OK;
}

// GR4HIP_IIR_AUTO: is float32 enough to carry this cascade's state through the parallel-in-time kernels?  Measured, not predicted: a three-tile noise vector through
// the kernels this handle would use, against the same cascade in float64 and in the reference's sequential float32 arithmetic (requested form) on the host.  More
// than 1e-5 of the output rms AND more than ten times the sequential float32 error: the handle runs GR4HIP_IIR_SEQUENTIAL_F32 instead.
static int iir_selftest(gr4hip_iir* f) {
    f->algo_in_use = GR4HIP_IIR_PARALLEL;
    if (f->algo == GR4HIP_IIR_SEQUENTIAL_F32) { f->algo_in_use = GR4HIP_IIR_SEQUENTIAL_F32; return GR4HIP_OK; }
    if (f->algo == GR4HIP_IIR_PARALLEL) return GR4HIP_OK;
    //  the verdict belongs to the cascade, not to the handle: one test per coefficient set and process.  A gain on a numerator does not change the conditioning
    // (every section's numerator enters the key divided by its largest coefficient), so the planner's regain -- a new handle per absorbed or cleared gain -- finds it here
    // instead of another upload / launch / download / host simulation with two device synchronisations.
    std::vector<uint32_t> key;
    bool                  cacheable = true; //  non-finite coefficients are not cached: there is nothing to learn from them twice
    {
        const IirSeqF32Coef& c = f->seq;
        int                  dev = 0;
        (void)hipGetDevice(&dev);
        key = {(uint32_t)c.form, (uint32_t)c.nsec, (uint32_t)c.nb, (uint32_t)c.na, (uint32_t)dev,
               (uint32_t)((dev_switch(kDevIirThreePass) ? 1 : 0) | (dev_switch(kDevIirLookback) ? 2 : 0) | (dev_switch(kDevIirNoSplit) ? 4 : 0))};
        const auto put = [&](float v) { uint32_t u; std::memcpy(&u, &v, 4); key.push_back(u); cacheable = cacheable && std::isfinite(v); };
        for (int s_ = 0; s_ < c.nsec; ++s_) {
            float mx = 0.f;
            for (int j = 0; j < c.nb; ++j) mx = std::max(mx, std::fabs(c.b[s_][j]));
            for (int j = 0; j < c.nb; ++j) put(mx > 0.f ? c.b[s_][j] / mx : 0.f);
            for (int j = 0; j < c.na; ++j) put(c.a[s_][j]);
        }
    }
    if (cacheable) {
        std::lock_guard<std::mutex> lk(g_selftest_mu);
        const auto it = g_selftest.find(key);
        if (it != g_selftest.end()) {
            f->algo_in_use = it->second.algo; f->selftest_parallel = it->second.e_par; f->selftest_f32 = it->second.e_f32;
            return GR4HIP_OK;
        }
    }
    // the test drives the handle's own state on the NULL stream: whatever a caller's streams still have in flight for this handle (gr4hip_iir_set_algo in
    // mid-stream) finishes first -- a settings change that measures something is a blocking call anyway (upload, launch, download, host simulation)
    GR4_HIP_TRY(hipDeviceSynchronize());
    const long         n = 3L * kIirBS * kIirL + 777;
    std::vector<float> x((size_t)n), y((size_t)n);
    uint32_t           lcg = 12345u;
    for (auto& v : x) { lcg = lcg * 1664525u + 1013904223u; v = (float)((int32_t)lcg) * (1.0f / 2147483648.0f); }
    DeviceBuffer dx, dy;
    int          rc = dx.ensure((size_t)n * sizeof(float));
    if (!rc) rc = dy.ensure((size_t)n * sizeof(float));
    if (rc) return rc;
    GR4_HIP_TRY(upload_fresh(dx.ptr, x.data(), (size_t)n * sizeof(float)));
    rc = iir_state_on(f, nullptr);
    if (!rc) rc = iir_process_parallel(f, static_cast<const float*>(dx.ptr), (size_t)n, static_cast<float*>(dy.ptr), nullptr);
    if (rc) return rc;
    GR4_HIP_TRY(hipMemcpy(y.data(), dy.ptr, (size_t)n * sizeof(float), hipMemcpyDeviceToHost));
    f->top = false;
    rc     = gr4hip_iir_reset(f); // the stream starts from zero state
    f->top = true;
    if (rc) return rc;
    const IirSeqF32Coef& c = f->seq;
    double ih64[kIirSeqMaxSec][kIirSeqMaxOrd + 1] = {}, oh64[kIirSeqMaxSec][kIirSeqMaxOrd + 1] = {};
    float  ih32[kIirSeqMaxSec][kIirSeqMaxOrd + 1] = {}, oh32[kIirSeqMaxSec][kIirSeqMaxOrd + 1] = {};
    double e_par = 0, e_f32 = 0, pw = 0;
    for (long i = 0; i < n; ++i) {
        double v64 = x[(size_t)i];
        float  v32 = x[(size_t)i];
        for (int s = 0; s < c.nsec; ++s) {
            v64 = iir_form_step<double>(GR4HIP_DF_II_TRANSPOSED, c.b[s], c.a[s], c.nb, c.na, ih64[s], oh64[s], v64); // (float64: any form; the transposed one carries the least state noise)
            v32 = iir_form_step<float>(c.form, c.b[s], c.a[s], c.nb, c.na, ih32[s], oh32[s], v32);
        }
        pw += v64 * v64;
        e_par = std::max(e_par, std::fabs((double)y[(size_t)i] - v64));
        e_f32 = std::max(e_f32, std::fabs((double)v32 - v64));
    }
    const double rms = std::sqrt(pw / (double)n);
    if (!(rms > 0.0) || !std::isfinite(rms)) return GR4HIP_OK; // nothing to judge (all-zero response, or an unstable filter: no arithmetic makes that one right)
    f->selftest_parallel = (float)(e_par / rms);
    f->selftest_f32      = (float)(e_f32 / rms);
    if (!(e_par / rms <= 1e-5) && !(e_par <= 10.0 * e_f32)) f->algo_in_use = GR4HIP_IIR_SEQUENTIAL_F32;
    if (cacheable) {
        std::lock_guard<std::mutex> lk(g_selftest_mu);
        if (g_selftest.size() >= kSelftestCacheMax) g_selftest.clear();
        g_selftest[key] = IirVerdict{f->algo_in_use, f->selftest_parallel, f->selftest_f32};
    }
    return GR4HIP_OK;
}

static int iir_take_error(gr4hip_iir* f, const char* where) {
    if (f->h_err && *f->h_err) {
        *f->h_err = 0;
        set_error("%s: the single-pass IIR kernel timed out in its look-back on an earlier call of this handle (the output of that call is invalid); "
                  "GR4HIP_IIR_THREE_PASS=1 selects the three-pass kernels", where);
        return GR4HIP_RUNTIME_ERROR;
    }
    return GR4HIP_OK;
}

// Block::reset() between two work() calls (Block.hpp:606, 1296): a host-side note; the state is zeroed on the stream of the next call, behind the launches that
// stream still has in flight for this handle (they write the other half of the state pair)
int gr4hip_iir_reset(gr4hip_iir_t* f) {
    GR4_REQUIRE(f, "iir_reset: null handle");
    if (f->top) f->zero_seq = true;
    if (f->part[0]) { int rc = gr4hip_iir_reset(f->part[0]); return rc ? rc : gr4hip_iir_reset(f->part[1]); }
    f->zero_state = true;
    return iir_take_error(f, "iir_reset");
}
static int iir_state_on(gr4hip_iir* f, hipStream_t st) {
    if (f->top && f->zero_seq && f->d_seq_state.ptr) {
        GR4_HIP_TRY(hipMemsetAsync(f->d_seq_state.ptr, 0, f->d_seq_state.bytes, st));
        f->zero_seq = false;
    }
    if (f->part[0]) { int rc = iir_state_on(f->part[0], st); return rc ? rc : iir_state_on(f->part[1], st); }
    if (f->zero_state) {
        for (int k = 0; k < 2; ++k) GR4_HIP_TRY(hipMemsetAsync(f->d_state[k].ptr, 0, kIirMaxM * sizeof(float), st));
        f->zero_state = false;
    }
    return GR4HIP_OK;
}

int gr4hip_iir_status(gr4hip_iir_t* f, gr4hip_stream_t stream) {
    GR4_REQUIRE(f, "iir_status: null handle");
    if (f->part[0]) { int rc = gr4hip_iir_status(f->part[0], stream); return rc ? rc : gr4hip_iir_status(f->part[1], stream); }
    GR4_HIP_TRY(hipStreamSynchronize(as_stream(stream)));
    return iir_take_error(f, "iir_status");
}

int gr4hip_iir_set_algo(gr4hip_iir_t* f, int algo) {
    GR4_REQUIRE(f && f->top, "iir_set_algo: null handle");
    GR4_REQUIRE(algo >= GR4HIP_IIR_AUTO && algo <= GR4HIP_IIR_SEQUENTIAL_F32, "iir_set_algo: unknown algo %d", algo);
    f->algo = algo;
    // the two evaluations keep different state (DF-II delay lines of the scan / the form's input and output histories): the filter restarts from zero state
    int rc = gr4hip_iir_reset(f);
    if (rc) return rc;
    return iir_selftest(f);
}

int gr4hip_iir_get_algo(const gr4hip_iir_t* f, int* algo_in_use, float* selftest_parallel, float* selftest_sequential_f32) {
    GR4_REQUIRE(f && algo_in_use, "iir_get_algo: null argument");
    *algo_in_use = f->algo_in_use;
    if (selftest_parallel) *selftest_parallel = f->selftest_parallel;
    if (selftest_sequential_f32) *selftest_sequential_f32 = f->selftest_f32;
    return GR4HIP_OK;
}

int gr4hip_iir_process(gr4hip_iir_t* f, const float* d_in, size_t n, float* d_out, gr4hip_stream_t stream) {
    GR4_REQUIRE(f, "iir_process: null handle");
    if (n == 0) return GR4HIP_OK;
    GR4_REQUIRE(d_in && d_out, "iir_process: null device pointer");
    if (const int rc = iir_state_on(f, as_stream(stream))) return rc; // a pending reset: onto this call's stream, in front of its launches
    if (f->top && f->algo_in_use == GR4HIP_IIR_SEQUENTIAL_F32) {
        hipLaunchKernelGGL(iir_sequential_kernel, dim3(1), dim3(64), 0, as_stream(stream), d_in, d_out, (long)n, f->seq, static_cast<float*>(f->d_seq_state.ptr));
        GR4_LAUNCH_CHECK();
        return GR4HIP_OK;
    }
    return iir_process_parallel(f, d_in, n, d_out, stream);
}

} // extern "C"

static int iir_process_parallel(gr4hip_iir_t* f, const float* d_in, size_t n, float* d_out, gr4hip_stream_t stream) {
    if (f->part[0]) {
        int rc = f->d_mid.ensure(n * sizeof(float));
        if (!rc) rc = iir_process_parallel(f->part[0], d_in, n, static_cast<float*>(f->d_mid.ptr), stream);
        return rc ? rc : iir_process_parallel(f->part[1], static_cast<const float*>(f->d_mid.ptr), n, d_out, stream);
    }
    hipStream_t st = as_stream(stream);
    const long  ln = (long)n;
    if (f->ord == 2) return f->nsec == 2 ? iir_run<2, 2>(f, d_in, d_out, ln, st) : f->nsec == 4 ? iir_run<2, 4>(f, d_in, d_out, ln, st) : iir_run<2, 8>(f, d_in, d_out, ln, st);
    return f->nsec == 1 ? iir_run<4, 1>(f, d_in, d_out, ln, st) : f->nsec == 2 ? iir_run<4, 2>(f, d_in, d_out, ln, st) : iir_run<4, 4>(f, d_in, d_out, ln, st);
}

// (library-internal, fir.hip: gr4hip_fir_iir_process) what the frequency-domain decimator needs to run this cascade as its store epilogue: up to four biquads on the
// parallel-in-time path, the per-section tables (coefficients + the 2 x 2 powers A^(14 2^k) of its chunk scan), the carried direct-form-II state (read from the
// current slot, written to the other: gr4hip_internal_iir_commit makes it current once the launch is known to stand) and the warm-up in blocks of 896 samples.
// Returns 0 when the cascade does not qualify (the caller keeps the two launches).
int gr4hip_internal_iir_fusable(gr4hip_iir* f, const float** d_tab, int* nsec, const float** d_state_in, float** d_state_out, int* warm_blocks, hipStream_t st) {
    if (!f || !f->top || f->part[0] || f->ord != 2 || f->algo_in_use != GR4HIP_IIR_PARALLEL || f->seq.nsec < 1 || f->seq.nsec > 4 || f->seq.nb > 3 || f->seq.na > 3) return 0;
    if (f->fuse_warm < 0) {
        f->fuse_warm = 0;
        std::vector<double> P = host_phi(f, 896), Pw = P;
        const int           M = f->M;
        for (int w = 1; w <= 4; ++w) {
            double nrm = 0;
            for (int i = 0; i < M; ++i) { double r = 0; for (int j = 0; j < M; ++j) r += std::fabs(Pw[i * M + j]); nrm = std::max(nrm, r); }
            if (nrm <= 1e-8) { f->fuse_warm = w; break; }
            std::vector<double> nx((size_t)M * M, 0.0);
            for (int i = 0; i < M; ++i)
                for (int k = 0; k < M; ++k)
                    for (int j = 0; j < M; ++j) nx[i * M + j] += Pw[i * M + k] * P[k * M + j];
            Pw = nx;
        }
        std::vector<float> tab((size_t)32 * f->seq.nsec, 0.f);
        for (int s = 0; s < f->seq.nsec; ++s) {
            float* t = tab.data() + 32 * s;
            t[0] = f->seq.b[s][0]; t[1] = f->seq.b[s][1]; t[2] = f->seq.b[s][2]; t[3] = f->seq.a[s][1]; t[4] = f->seq.a[s][2];
            double A[4] = {-(double)t[3], -(double)t[4], 1.0, 0.0}, R[4] = {1, 0, 0, 1};
            const auto mm = [](const double* x, const double* y, double* o) { const double r[4] = {x[0] * y[0] + x[1] * y[2], x[0] * y[1] + x[1] * y[3], x[2] * y[0] + x[3] * y[2], x[2] * y[1] + x[3] * y[3]}; std::memcpy(o, r, sizeof(r)); };
            for (int i = 0; i < 14; ++i) mm(R, A, R); // A^14: one chunk
            for (int k = 0; k < 5; ++k) {
                for (int i = 0; i < 4; ++i) t[8 + 4 * k + i] = (float)R[i];
                mm(R, R, R);
            }
        }
        if (f->d_fuse_tab.ensure(tab.size() * sizeof(float)) != GR4HIP_OK || upload_fresh(f->d_fuse_tab.ptr, tab.data(), tab.size() * sizeof(float)) != hipSuccess) f->fuse_warm = 0;
    }
    if (f->fuse_warm <= 0) return 0;
    if (iir_state_on(f, st)) return 0; // (a pending reset goes in front of the launch that reads the state)
    *d_tab       = static_cast<const float*>(f->d_fuse_tab.ptr);
    *nsec        = f->seq.nsec;
    *d_state_in  = static_cast<const float*>(f->d_state[f->cur].ptr);
    *d_state_out = static_cast<float*>(f->d_state[f->cur ^ 1].ptr);
    *warm_blocks = f->fuse_warm;
    return 1;
}
void gr4hip_internal_iir_commit(gr4hip_iir* f) { f->cur ^= 1; }

extern "C" {

int gr4hip_iir_destroy(gr4hip_iir_t* f) { delete f; return GR4HIP_OK; }

} // extern "C"
