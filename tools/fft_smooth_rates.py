#!/usr/bin/env python
"""developer tool: FFT block |X|^2 rates at {2,3,5}-smooth sizes (mixed-radix kernel, fft_smooth.hpp) beside their power-of-two neighbours"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnuradio4_amd as G
out = []
import os
SIZES = [int(v) for v in os.environ.get('SMOOTH_SIZES', '1000,1024,1536,2048,3000,4096,6000,8000,8192,1009').split(',')]
for N in SIZES:
    n = (1 << 26) // N * N
    xc = G.synth_c32(n)
    m2 = torch.empty(n, dtype=torch.float32, device="cuda")
    F = G.FFT(N, "Hann")
    for _ in range(3):
        F.mag2(xc, m2.view(n // N, N))
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        F.mag2(xc, m2.view(n // N, N))
    b.record(); b.synchronize()
    out.append("%d: %.0f" % (N, n * 10 / a.elapsed_time(b) / 1e6))
print("FFT mag2 Gsamples/s  " + "  ".join(out))
