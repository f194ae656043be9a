// NOT BUILT, NOT SHIPPED: the wave-specialised form of csrc/fir_decim_f16.hip's kernel as it was measured in round 5 (profiles/r05_decim_bounds.txt: parity-green, - 6 %).
// It lived inside fir_decim_f16.hip between the kernel and fir_decim_f16_make_table, used that file's dh_contract<D, KQ, PL, MIDBAR> (a __syncthreads() after fragment NM / 3 of the
// product loop when MIDBAR) and was launched as <<<ceil(nseg / spw), 256 + 64 NPW, 2 * lds + 2 * 4 * (32 / D) * 64 * 4 * sizeof(float)>>> under GR4HIP_DH_WS=1.
// ---------------------------------------------------------------------------------------------------------------------------------------------------------------------------
// (round 5) the same arithmetic with the waves SPECIALISED: one 512-lane workgroup per CU, waves 0 - 3 ("consumers", one per SIMD) hold the tap fragments and do nothing but the
// products of segment t -- planes[t & 1] -> part[t & 1] --, waves 4 - 7 ("producers", one per SIMD beside a consumer) do everything else during the same step: take segment t - 1
// out (part[(t - 1) & 1] -> y, its output powers, its verdict), and stage segment t + 1 (statistics, block exponent, two-term split -> planes[(t + 1) & 1]) from registers that
// were loaded three steps earlier.  Two barriers per step: B1 when the producers' statistics words are complete (the consumers pass it a third of the way through their products),
// B2 at the step's end.  Why: in the kernel above the four phases of a segment run one after the other in every wave, two workgroups per CU overlap them only partly, and stream and
// arithmetic ADD UP (profiles/r05_decim_bounds.txt: no loads + 44 %, no products + 14 %, no statistics + 14 %); here the matrix pipe, the vector ALU and the HBM stream of one CU work
// on three different segments at once.  LDS: two plane pairs (75 KB) + two partial-tile sets: 107 KB at D = 8.
template <int D, int KQ, int HOOK, int NPW /*producer waves: 4 or 8 (one or two per SIMD beside the SIMD's consumer)*/>
__global__ __launch_bounds__(256 + 64 * NPW, 1) void fir_decim_f16x2_ws_kernel(const float* __restrict__ x, const float* __restrict__ hist /*hist[h] = x[-Kh + h]*/, int Kh, const unsigned short* __restrict__ tab,
                                                                     float* __restrict__ y, long n_out, long n_in, float* __restrict__ new_hist, int guard, int seg_per_wg,
                                                                     unsigned char* __restrict__ flags, int cplx, BdHooks hk) {
    constexpr int TR = 32 / D, SO = kDhSegIn / D, Hb = 128 * KQ - 16 * D, NS = kDhSegIn + Hb;
    static_assert(Hb > 0, "the window must hold a tile's 16 D input samples");
    constexpr int PL  = NS + 8 * (NS / 512 + 1) + 16;
    constexpr int NPT = 64 * NPW;                    // producer lanes
    constexpr int NL4 = (NS / 4 + NPT - 1) / NPT;
    constexpr int kPartFloats = 4 * TR * 64 * 4;
    const u32x4_h* afrag = reinterpret_cast<const u32x4_h*>(tab);
    const float    inv_t = *reinterpret_cast<const float*>(tab + dh_frag_units(KQ));
    const float    gthr  = *reinterpret_cast<const float*>(tab + dh_frag_units(KQ) + 4);
    extern __shared__ __attribute__((aligned(16))) unsigned short ws_lds[]; // [2 buffers][2 planes][PL] f16, then [2][4][TR][64][4] float partial tiles
    __shared__ __attribute__((aligned(16))) unsigned stat[3 * NPW];
    __shared__ __attribute__((aligned(16))) float    ystat[4][16];
    __shared__ int kindv[2]; // the class of the segment in planes[b]: 0 = the consumers evaluate it
    float* const partb = reinterpret_cast<float*>(ws_lds + 4 * PL);
    const bool producer = __builtin_amdgcn_readfirstlane((int)threadIdx.x) >= 256; // (a scalar: the two roles are two loops with the same barrier count)
    const int  tid = producer ? (int)threadIdx.x - 256 : (int)threadIdx.x, lane = tid & 63, wave = tid >> 6, col = lane & 15, kq = lane >> 4; // (role-local)
    const long nseg = (n_out + SO - 1) / SO, sfirst = (long)blockIdx.x * seg_per_wg, slast = sfirst + seg_per_wg < nseg ? sfirst + seg_per_wg : nseg;
    if (sfirst >= slast) return;
    const int n = (int)(slast - sfirst);
    if (!producer) {
        u32x4_h a[2][KQ];
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int ks = 0; ks < KQ; ++ks) a[p][ks] = afrag[((wave * 3 + p) * KQ + ks) * 64 + lane];
        for (int t = -1; t <= n; ++t) {
            const int b = t & 1;
            if (t >= 0 && t < n && kindv[b] == 0)
                dh_contract<D, KQ, PL, true>(a, ws_lds + 2 * PL * b, reinterpret_cast<float(*)[TR][64][4]>(partb + kPartFloats * b), wave, lane); // (B1 inside)
            else
                __syncthreads(); // B1
            __syncthreads();     // B2
        }
        return;
    }
    // ---- producers
    auto P  = [](int s_) { return s_ + 8 * (s_ >> 9); };
    auto xs = [&](long i) __attribute__((always_inline)) -> float { return i >= 0 ? (i < n_in ? x[i] : 0.f) : (i >= -(long)Kh ? hist[Kh + i] : 0.f); };
    using seg_regs = float4[NL4];
    auto load_next = [&](seg_regs& nxt, long sg) __attribute__((always_inline)) { // sg >= 1: nothing below 0
        const long   i0   = sg * kDhSegIn - Hb;
        const long   nrec = n_in - i0 < (long)NS ? n_in - i0 : (long)NS;
        const rsrc_t r    = make_rsrc(x + i0, (unsigned)(nrec > 0 ? nrec * 4 : 0));
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, tid * 16, NPT * u * 16, 0);
            nxt[u]       = make_float4(__uint_as_float(v[0]), __uint_as_float(v[1]), __uint_as_float(v[2]), __uint_as_float(v[3]));
        }
    };
    auto load_seg = [&](seg_regs& nxt, long sg) __attribute__((always_inline)) {
        if (sg > 0) load_next(nxt, sg);
        else {
#pragma unroll
            for (int u = 0; u < NL4; ++u) {
                const int  q  = tid + NPT * u;
                const long i0 = -(long)Hb + 4L * q;
                nxt[u] = q < NS / 4 ? make_float4(xs(i0), xs(i0 + 1), xs(i0 + 2), xs(i0 + 3)) : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };
    auto hook_loaded = [&](seg_regs& nxt, long sg) __attribute__((always_inline)) { // (fir_decim_f16x2_kernel's, on the producers' 256 lanes)
        if constexpr (HOOK == 2) {
            BdRotor rot = bd_rotor_start(hk.pre, sg * kDhSegIn - Hb, tid);
            rot.step    = (unsigned long long)(2 * NPT) * rot.inc; // (a lane's next float4 is NPT float4s = 2 NPT complex samples further on)
#pragma unroll
            for (int u = 0; u < NL4; ++u) {
                const float4 v = nxt[u];
                float4       w = bd_rotor_next(v, rot);
                if (sg == 0) {
                    const long fi = 4L * (tid + NPT * u) - Hb;
                    if (fi < 0) { w.x = v.x; w.y = v.y; }
                    if (fi + 2 < 0) { w.z = v.z; w.w = v.w; }
                }
                nxt[u] = w;
            }
        } else if constexpr (HOOK == 1) {
            if (hk.pre.n_ops > 0) {
#pragma unroll
                for (int u = 0; u < NL4; ++u) {
                    const int  q  = tid + NPT * u;
                    const long fi = sg * kDhSegIn - Hb + 4L * q;
                    if (q < NS / 4 && fi + 3 >= 0 && fi < n_in) {
                        const float4 w = bd_hook4(nxt[u], hk.pre, cplx, fi);
                        float4       v = nxt[u];
                        if (fi >= 0) v = w;
                        else if (fi + 2 >= 0) { v.z = w.z; v.w = w.w; if (!cplx && fi + 1 >= 0) v.y = w.y; }
                        else if (!cplx) v.w = w.w;
                        if (fi + 3 >= n_in) {
                            if (fi + 1 >= n_in) v.y = 0.f;
                            if (fi + 2 >= n_in) v.z = 0.f;
                            v.w = 0.f;
                        }
                        nxt[u] = v;
                    }
                }
            }
        }
    };
    auto put_stats = [&](const seg_regs& nxt) __attribute__((always_inline)) {
        float    mf = 0.f, px = 0.f;
        unsigned mn = 0xffffffffu;
#pragma unroll
        for (int u = 0; u < NL4; ++u) {
            const float m4 = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(nxt[u].x), __builtin_fabsf(nxt[u].y)), __builtin_fmaxf(__builtin_fabsf(nxt[u].z), __builtin_fabsf(nxt[u].w)));
            mf = __builtin_fmaxf(mf, m4);
            mn = min(mn, __float_as_uint(m4) - 1u);
            px = fmaf(nxt[u].x, nxt[u].x, fmaf(nxt[u].y, nxt[u].y, fmaf(nxt[u].z, nxt[u].z, fmaf(nxt[u].w, nxt[u].w, px))));
        }
        unsigned mx = __float_as_uint(mf);
        mx = hf_wave_reduce_u32(mx, [](unsigned a_, unsigned b_) { return a_ > b_ ? a_ : b_; });
        mn = hf_wave_reduce_u32(mn, [](unsigned a_, unsigned b_) { return a_ < b_ ? a_ : b_; });
        px = hf_wave_sum(px);
        if (lane == 0) { stat[wave] = mx; stat[NPW + wave] = mn; stat[2 * NPW + wave] = __float_as_uint(px); }
    };
    auto block_scale = [&](float& s, float& inv_s, float& px) __attribute__((always_inline)) -> int {
        unsigned mxv = 0u, mnv = 0xffffffffu;
        px = 0.f;
#pragma unroll
        for (int w4 = 0; w4 < NPW; w4 += 4) { // (a fixed order: every lane of every producer wave adds the same numbers the same way)
            const uint4 m4 = *reinterpret_cast<const uint4*>(&stat[w4]), n4 = *reinterpret_cast<const uint4*>(&stat[NPW + w4]), p4 = *reinterpret_cast<const uint4*>(&stat[2 * NPW + w4]);
            px += (__uint_as_float(p4.x) + __uint_as_float(p4.y)) + (__uint_as_float(p4.z) + __uint_as_float(p4.w));
            mxv = max(mxv, max(max(m4.x, m4.y), max(m4.z, m4.w)));
            mnv = min(mnv, min(min(n4.x, n4.y), min(n4.z, n4.w)));
        }
        const unsigned mx = __builtin_amdgcn_readfirstlane(mxv), mn = __builtin_amdgcn_readfirstlane(mnv);
        const int e = (int)(mx >> 23), el = (int)(mn >> 23);
        const int slow = (e == 255 || px != px) ? 2 : ((mn != 0xffffffffu && e - el > kHfMaxRange) ? 1 : 0);
        const int ec = e < 15 ? 15 : (e > 254 ? 254 : e);
        s     = __uint_as_float((unsigned)(268 - ec) << 23);
        inv_s = __uint_as_float((unsigned)(ec - 14) << 23);
        return slow;
    };
    auto take_out = [&](long sg, float k, float& py, const float* __restrict__ part_) __attribute__((always_inline)) {
        auto part = reinterpret_cast<const float(*)[TR][64][4]>(part_);
#pragma unroll
        for (int tr = wave; tr < TR; tr += 4) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) v[r] = ((part[0][tr][lane][r] + part[1][tr][lane][r]) + (part[2][tr][lane][r] + part[3][tr][lane][r])) * k;
            const long o = sg * SO + (long)(16 * TR) * col + 16 * tr + 4 * kq;
#pragma unroll
            for (int r = 0; r < 4; ++r)
                if (o + r < n_out) py = fmaf(v[r], v[r], py);
            if constexpr (HOOK == 1) {
                if (hk.post.n_ops > 0) { const float4 w = bd_hook4(make_float4(v[0], v[1], v[2], v[3]), hk.post, cplx, o); v[0] = w.x; v[1] = w.y; v[2] = w.z; v[3] = w.w; }
            }
            if (o + 3 < n_out) *reinterpret_cast<float4*>(y + o) = make_float4(v[0], v[1], v[2], v[3]);
            else {
                for (int r = 0; r < 4; ++r)
                    if (o + r < n_out) y[o + r] = v[r];
            }
        }
    };
    auto rejected = [&](float px) __attribute__((always_inline)) -> bool {
        const float pc = (ystat[0][col] + ystat[1][col]) + (ystat[2][col] + ystat[3][col]);
        return __builtin_amdgcn_readfirstlane((int)(16.f * hf_row_min(pc) * (float)D < gthr * px)) != 0;
    };
    for (int i = tid; i < n; i += NPT) flags[sfirst + i] = 0;
    // the state of the two segments behind the one being staged: (k0, p0, i0) segment t (in the consumers' hands during step t), (k1, p1, i1) segment t - 1 (taken out in step t)
    int   k0 = -1, k1 = -1;
    float p0 = 0.f, p1 = 0.f, i0 = 0.f, i1 = 0.f;
    // step t with the registers that hold segment t + 1
    auto step = [&](seg_regs& nxt, int t) __attribute__((always_inline)) {
        float py = 0.f;
        if (k1 == 0 && wave < 4) take_out(sfirst + t - 1, inv_t * i1, py, partb + kPartFloats * ((t - 1) & 1));
        if (wave >= TR || wave >= 4) py = 0.f;
        else if (k1 != 0 || (sfirst + t - 1) * SO + (long)(16 * TR) * col >= n_out) py = __builtin_inff();
        py = hf_column_sum(py);
        if (lane < 16 && wave < 4) ystat[wave][lane] = py;
        const bool stage = t + 1 < n;
        if (stage) {
            hook_loaded(nxt, sfirst + t + 1);
            put_stats(nxt);
        }
        __syncthreads(); // B1: the statistics words and the output powers are complete
        if (guard && k1 == 0 && rejected(p1) && tid == 0) flags[sfirst + t - 1] = 3;
        int   kn = -1;
        float pn = 0.f, in = 0.f;
        if (stage) {
            float s;
            kn = block_scale(s, in, pn);
            unsigned short* pls = ws_lds + 2 * PL * ((t + 1) & 1);
#pragma unroll
            for (int u = 0; u < NL4; ++u) {
                const int q = tid + NPT * u;
                if (NPT * (u + 1) <= NS / 4 || q < NS / 4) {
                    unsigned h0, l0, h1, l1;
                    hf_split2(nxt[u].x, nxt[u].y, s, h0, l0);
                    hf_split2(nxt[u].z, nxt[u].w, s, h1, l1);
                    *reinterpret_cast<uint2*>(pls + P(4 * q))      = make_uint2(h0, h1);
                    *reinterpret_cast<uint2*>(pls + PL + P(4 * q)) = make_uint2(l0, l1);
                }
            }
            if (tid == 0) {
                kindv[(t + 1) & 1] = kn;
                if (kn != 0) flags[sfirst + t + 1] = (unsigned char)kn;
            }
            if (t + 4 < n) load_next(nxt, sfirst + t + 4);
        }
        __syncthreads(); // B2
        k1 = k0; p1 = p0; i1 = i0;
        k0 = kn; p0 = pn; i0 = in;
    };
    {
        float4 r0[NL4], r1[NL4], r2[NL4];
        load_seg(r0, sfirst);
        if (n > 1) load_next(r1, sfirst + 1);
        if (n > 2) load_next(r2, sfirst + 2);
        for (int t = -1; t <= n; t += 3) {
            step(r0, t);
            if (t + 1 <= n) step(r1, t + 1);
            if (t + 2 <= n) step(r2, t + 2);
        }
    }
    if (new_hist != nullptr && blockIdx.x == 0 && tid < 256) bd_new_hist<HOOK != 0>(x, hist, Kh, n_in, new_hist, tid, hk);
}

