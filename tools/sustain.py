#!/usr/bin/env python
"""developer tool: the headline chain launched back to back for a few seconds, rate per group of 20 launches (does the rate hold once the package sits at its power cap?)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnuradio4_amd as G
k = np.arange(256); taps = np.hamming(256) * 0.2 * np.sinc(0.2 * (k - 127.5)); taps = (taps / taps.sum()).astype(np.float32)
n = 1 << 28
nbuf = int(os.environ.get("SUSTAIN_BUFFERS", "1"))
xs = [G.synth_c32(n, seed=42 + i) for i in range(nbuf)]
ms = [torch.empty(n, dtype=torch.float32, device="cuda") for _ in range(nbuf)]
c = G.Chain(taps, 8192, "None", int(os.environ.get("SUSTAIN_ALGO", "0")))
groups = int(os.environ.get("SUSTAIN_GROUPS", "30"))
evs = [torch.cuda.Event(enable_timing=True) for _ in range(groups + 1)]
for i in range(3): c.process_bulk(xs[0], ms[0])
torch.cuda.synchronize()
evs[0].record()
for g in range(groups):
    for r in range(20): c.process_bulk(xs[r % nbuf], ms[r % nbuf])
    evs[g + 1].record()
torch.cuda.synchronize()
rates = [n * 20 / (evs[g].elapsed_time(evs[g + 1]) * 1e-3) / 1e9 for g in range(groups)]
print("buffers %d: Gsamples/s per 20 launches:" % nbuf, " ".join("%.0f" % r for r in rates))
