#!/usr/bin/env python
"""developer tool: interpolating FIR rates (input Gsamples/s, useful TFLOP/s, HBM TB/s)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np, torch
import gnuradio4_amd as G
from _timing import steady
for dtype, S in ((torch.float32, 1), (torch.complex64, 2)):
    for L, K in ((2, 64), (2, 256), (3, 91), (3, 256), (5, 100), (5, 320), (6, 96), (6, 384), (4, 256), (8, 256), (8, 1024), (16, 256), (7, 128), (12, 256), (32, 512)):
        n = (1 << 26) // (L * S)
        x = G.synth_f32(n) if S == 1 else G.synth_c32(n)
        b = (np.hamming(K) / K).astype(np.float32)
        f = G.fir_interpolator(b, L, dtype)
        out = torch.empty(n * L, dtype=dtype, device="cuda")
        ms = steady(lambda: f.process_bulk(x, out)) * 1e3
        flop = 2.0 * K / L * n * L * S
        print("%-9s L=%d K=%4d: %7.2f G input samples/s  %7.2f G output samples/s  %6.2f TFLOP/s  %5.2f TB/s" % ("float32" if S == 1 else "complex64", L, K, n / ms / 1e6, n * L / ms / 1e6, flop / ms / 1e9, n * (1 + L) * 4 * S / ms / 1e9))
