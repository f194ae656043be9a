#!/usr/bin/env python
"""developer tool: where a +Inf, a NaN and a 3.4e38 sample end up in the FIR / chain outputs (default algorithm and GR4HIP_FIR_EXACT_F32) against the oracle"""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import gnuradio4_amd as G
import oracle_lib as O

def runs(mask):
    idx = np.flatnonzero(mask)
    if len(idx) == 0: return []
    cuts = np.flatnonzero(np.diff(idx) > 1)
    starts = np.r_[idx[0], idx[cuts + 1]]; ends = np.r_[idx[cuts], idx[-1]]
    return list(zip(starts.tolist(), ends.tolist()))

def aligned(x):
    t = torch.empty(x.size + 4, dtype=torch.from_numpy(x[:1]).dtype, device="cuda")[4 if x.dtype == np.float32 else 2:][:x.size]
    t.copy_(torch.from_numpy(x)); return t

n = 300_000
pos = {"inf": 50_001, "nan": 120_003, "big": 200_005}
for cplx, ntaps in ((False, 64), (False, 200), (True, 256), (True, 64)):
    b = O.design_taps_hamming_lowpass(ntaps, 0.1)
    x = (O.signal_c32 if cplx else O.signal_f32)(7, n)
    x[pos["inf"]] = np.inf; x[pos["nan"]] = np.nan; x[pos["big"]] = 3.4e38
    truth, _ = O.fir(b, x)
    t32 = truth.astype(np.complex64 if cplx else np.float32)
    print(f"== {'complex' if cplx else 'float'} {ntaps} taps: oracle non-finite runs {runs(~np.isfinite(t32))}")
    for name, algo in (("default", G.capi.FIR_AUTO), ("exact_f32", G.capi.FIR_EXACT_F32)):
        f = G.fir_filter(b, torch.complex64 if cplx else torch.float32)
        f.set_algo(algo)
        y = f.process_bulk(aligned(x)).cpu().numpy()
        bad = ~np.isfinite(y)
        same_class = np.array_equal(np.isnan(y.real), np.isnan(t32.real)) and np.array_equal(np.isposinf(y.real), np.isposinf(t32.real)) and np.array_equal(np.isneginf(y.real), np.isneginf(t32.real))
        ok = np.isfinite(t32) & ~bad
        err = np.max(np.abs(y[ok] - truth[ok])) / np.sqrt(np.mean(np.abs(truth[ok]) ** 2))
        print(f"   {name:9s}: non-finite runs {runs(bad)}  classes identical to the oracle: {same_class}  err elsewhere {err:.2e}  finite where the oracle is not: {int(np.sum(~np.isfinite(t32) & ~bad))}")
# the chain
N = 8192
for ntaps in (256, 64):
    b = O.design_taps_hamming_lowpass(ntaps, 0.1)
    x = O.signal_c32(9, 12 * N)
    x[2 * N + 100] = np.inf; x[5 * N - 100] = np.nan; x[8 * N - 200] = 3.4e38; x[10 * N + 5] = np.nan
    truth, _ = O.chain(b, x, N, 0, truth=True)
    tb = ~np.isfinite(truth.reshape(-1, N).astype(np.float32))
    print(f"== chain {ntaps} taps: oracle frames with non-finite bins {[(i, int(r.sum())) for i, r in enumerate(tb) if r.any()]}")
    for name, algo in (("auto", 0), ("fused_fd", 3), ("time_domain", 4)):
        ch = G.Chain(b, N, "None", algo)
        y = ch.process_bulk(torch.from_numpy(x).cuda()).cpu().numpy().reshape(-1, N)
        gb = ~np.isfinite(y)
        okf = [i for i in range(12) if not tb[i].any() and not gb[i].any()]
        t2 = truth.reshape(-1, N)
        err = max(float(np.max(np.abs(y[i] - t2[i]) / np.maximum(np.abs(t2[i]), np.sqrt(np.mean(t2[i] ** 2))))) for i in okf)
        print(f"   {name:11s}: frames with non-finite bins {[(i, int(r.sum())) for i, r in enumerate(gb) if r.any()]}  err on clean frames {err:.2e}  ratio/td {ch.last_power_ratio()}")
