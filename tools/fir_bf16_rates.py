#!/usr/bin/env python
"""developer tool: fir_filter<float> 65 .. 256 taps: the bf16 three-term kernel (default) beside the f32 MFMA kernel (GR4HIP_FIR_NO_BF16X3=1 in a second run)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from _timing import steady
import gnuradio4_amd as G
n = 1 << 27
x = G.synth_f32(n); y = torch.empty_like(x)
for K in (32, 33, 48, 64, 65, 81, 82, 128, 145, 146, 200, 256):
    b = (np.hamming(K) / K).astype(np.float32)
    f = G.fir_filter(b, torch.float32)
    t = steady(lambda: f.process_bulk(x, y))
    print("%4d taps: %7.1f Gsamples/s = %5.2f TB/s  (%6.1f TFLOP/s float32-equivalent)" % (K, n / t / 1e9, 8.0 * n / t / 1e12, 2.0 * K * n / t / 1e12))
