#!/usr/bin/env python
"""developer tool: one tone_ratio.py case of a family (cplx, D, taps, amp) under every FIR algo: error vs float64 and vs the reference float32 sum.
usage: diag_family.py cplx D taps amp"""
import sys
sys.path.insert(0, "."); sys.path.insert(0, "tests")
import numpy as np, torch
import gnuradio4_amd as G
import oracle_lib as O
from gnuradio4_amd import capi
cplx, D, nt, amp = bool(int(sys.argv[1])), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
rng = np.random.default_rng(5)
n = ((1 << 18) // (4 * D)) * 4 * D
k = np.arange(nt); fc = 0.4 / D if D > 1 else 0.05
taps = (np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2.0))).astype(np.float32)
def rel(y, t, sl):
    y, t = y[sl], t[sl]; rms = float(np.sqrt(np.mean(np.abs(t) ** 2)))
    return float(np.max(np.abs(y - t) / np.maximum(np.abs(t), rms)))
for fq in (0.21, 0.3, 0.44):
    ph = 2 * np.pi * fq * np.arange(n)
    x = rng.standard_normal(n) + (1j * rng.standard_normal(n) if cplx else 0)
    x = (x * 0.05 + amp * (np.exp(1j * ph) if cplx else np.cos(ph))).astype(np.complex64 if cplx else np.float32)
    truth = O.fir(taps, x, acc64=True)[0][::D]; ref32 = O.fir(taps, x, acc64=False)[0][::D]
    sl = slice(nt // D + 1, None)
    print(f"fq {fq}: rms(out) {np.sqrt(np.mean(np.abs(truth[sl])**2)):.3e}  e32 {rel(ref32, truth, sl):.2e}", end="")
    for name in ("FIR_AUTO", "FIR_TIME_DOMAIN", "FIR_EXACT_F32"):
        f = G.fir_filter(taps, torch.complex64 if cplx else torch.float32, decimate=D)
        f.set_algo(getattr(capi, name))
        pad = 2 if cplx else 4
        t = torch.empty(len(x) + pad, dtype=torch.complex64 if cplx else torch.float32, device="cuda")[pad:]
        t.copy_(torch.from_numpy(x))
        y = f.process_bulk(t).cpu().numpy()
        print(f"  {name} {rel(y, truth, sl):.2e}", end="")
    print()
