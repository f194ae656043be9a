#!/usr/bin/env python
"""developer tool: |X|^2 rate of the FFT block at every {2,3,5}-smooth size <= 8192 that has a compile-time plan; JSON {N: Gsamples/s} to stdout"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnuradio4_amd as G
sizes = sorted({2 ** a * 3 ** b * 5 ** c for a in range(14) for b in range(9) for c in range(6)} - {2 ** a for a in range(14)})
sizes = [n for n in sizes if 18 <= n <= 8192]
n_tot = 1 << 25
xc_all = G.synth_c32(n_tot)
m2_all = torch.empty(n_tot, dtype=torch.float32, device="cuda")
res = {}
for N in sizes:
    n = n_tot // N * N
    xc, m2 = xc_all[:n], m2_all[:n].view(n // N, N)
    F = G.FFT(N, "Hann")
    for _ in range(3):
        F.mag2(xc, m2)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(6):
        F.mag2(xc, m2)
    b.record(); b.synchronize()
    res[N] = round(n * 6 / a.elapsed_time(b) / 1e6, 1)
print(json.dumps(res))
