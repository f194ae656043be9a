#!/usr/bin/env python
"""developer tool: |FFT_8192|^2 (no window, no filter) on the frame pipeline kernels, Gs/s"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnuradio4_amd as G
n = 1 << 28
xc = G.synth_c32(n)
m2 = torch.empty(n, dtype=torch.float32, device="cuda")
F = G.FFT(8192, "None")
for _ in range(5):
    F.mag2(xc, m2.view(n // 8192, 8192))
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10):
    F.mag2(xc, m2.view(n // 8192, 8192))
b.record(); b.synchronize()
print("FFT8192 mag2 (GR4HIP_CHAIN16=%s): %.1f Gs/s" % (os.environ.get("GR4HIP_CHAIN16", "0"), n * 10 / a.elapsed_time(b) / 1e6))
