#!/usr/bin/env python
"""developer tool: single-pass IIR (look-back) against the three-pass kernels on random span lengths / chunkings, many repetitions (race hunting)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnuradio4_amd as G
from gnuradio4_amd import capi
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
b4, a4 = G.blocks.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH)
cases = {"biquad4": (b4, a4), "pole1": (np.array([[0.3]], np.float32), np.array([[1.0, -0.7]], np.float32)),
         "order4x2": (np.array([[0.1, 0.2, 0.3, 0.2, 0.1]] * 2, np.float32), np.array([[1.0, -0.9, 0.5, -0.1, 0.02]] * 2, np.float32))}
worst = 0.0
for it in range(int(sys.argv[2]) if len(sys.argv) > 2 else 40):
    n = int(rng.integers(1, 1 << 24))
    x = G.synth_f32(n, seed=int(rng.integers(1, 1000)))
    cuts = sorted(set([0, n] + [int(c) for c in rng.integers(0, n, size=int(rng.integers(0, 4)))]))
    name = list(cases)[it % len(cases)]
    b, a = cases[name]
    ys = {}
    for mode in ("one", "three"):
        if mode == "three":
            capi.developer_switch("GR4HIP_IIR_THREE_PASS", 1)
        else:
            capi.developer_switch("GR4HIP_IIR_THREE_PASS", 0)
        f = G.iir_filter(b, a)
        y = torch.empty_like(x)
        for lo, hi in zip(cuts[:-1], cuts[1:]):
            f.process_bulk(x[lo:hi], y[lo:hi])
        ys[mode] = y.double()
    rms = float(ys["three"].pow(2).mean().sqrt()) or 1.0
    err = float((ys["one"] - ys["three"]).abs().max()) / rms
    worst = max(worst, err)
    if err > 2e-5:
        print("MISMATCH", name, n, cuts, err)
        sys.exit(1)
print("iir stress ok, worst rel diff %.3g" % worst)
