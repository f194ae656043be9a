#!/usr/bin/env python
"""developer tool: decimating fir_filter<complex<float>> (real taps) and short-tap decimating fir_filter<float> over (decimation, taps): input Gsamples/s"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from _timing import steady
import gnuradio4_amd as G
n = 1 << 26
xc = torch.view_as_complex(G.synth_f32(2 * n).view(-1, 2))
xf = G.synth_f32(n)
for dt, x, bpi in ((torch.complex64, xc, 8), (torch.float32, xf, 4)):
    for D, K in ((2, 16), (2, 64), (3, 24), (3, 96), (4, 32), (4, 64), (4, 128), (8, 32), (8, 64), (8, 128), (8, 256), (10, 80), (16, 64), (16, 128), (16, 256), (32, 256)):
        b = (np.hamming(K) / K).astype(np.float32)
        f = G.fir_filter(b, dt, decimate=D)
        nn = n - n % D
        y = torch.empty(nn // D, dtype=dt, device="cuda")
        t = steady(lambda: f.process_bulk(x[:nn], y))
        print("%-9s D=%2d K=%4d: %7.1f G input samples/s  %5.2f TB/s" % (str(dt).split(".")[1], D, K, nn / t / 1e9, nn * bpi * (1 + 1.0 / D) / t / 1e12), flush=True)
