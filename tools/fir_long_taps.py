#!/usr/bin/env python
"""developer tool: fir_filter<float | complex<float>> beyond 256 taps (where the fixed-size matrix-pipe / fast-convolution kernels stop)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from _timing import steady
import gnuradio4_amd as G
n = 1 << 26
xf = G.synth_f32(n); xc = G.synth_c32(n // 2)
for K in (256, 257, 512, 1024, 2048, 4096):
    b = (np.hamming(K) / K).astype(np.float32)
    f = G.fir_filter(b, torch.float32); y = torch.empty_like(xf)
    t = steady(lambda: f.process_bulk(xf, y))
    g = G.fir_filter(b, torch.complex64); yc = torch.empty_like(xc)
    tc = steady(lambda: g.process_bulk(xc, yc))
    print("%4d taps: float %7.1f Gsamples/s = %6.1f TFLOP/s | complex %7.1f Gsamples/s = %6.1f TFLOP/s" % (K, n / t / 1e9, 2.0 * K * n / t / 1e12, n / 2 / tc / 1e9, 4.0 * K * n / 2 / tc / 1e12))
