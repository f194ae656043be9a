#!/usr/bin/env python
"""developer tool: FFT block |X|^2 rates at a few sizes (A/B runs with tools/ab_libs-style library swaps)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnuradio4_amd as G
n = 1 << 27
xc = G.synth_c32(n)
m2 = torch.empty(n, dtype=torch.float32, device="cuda")
out = []
for N in (256, 1024, 4096, 8192):
    F = G.FFT(N, "Hann")
    for _ in range(3):
        F.mag2(xc, m2.view(n // N, N))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        F.mag2(xc, m2.view(n // N, N))
    b.record(); b.synchronize()
    out.append("%d: %.0f" % (N, n * 10 / a.elapsed_time(b) / 1e6))
print("FFT mag2 Gs/s  " + "  ".join(out))
