#!/usr/bin/env python
"""developer tool: decimating fir_filter<float> over (decimation, taps): input Gsamples/s and the direct-form-equivalent TFLOP/s"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from _timing import steady
import gnuradio4_amd as G
n = 1 << 27
x = G.synth_f32(n)
for D, K in ((2, 64), (2, 128), (2, 256), (3, 240), (4, 220), (5, 210), (9, 150), (8, 128), (8, 167), (3, 96), (4, 128), (4, 512), (5, 160), (8, 256), (8, 1024), (9, 288), (10, 320), (11, 352), (12, 384), (16, 512), (20, 640), (25, 800), (48, 1536), (100, 1600), (16, 2048), (32, 1024), (64, 2048)):
    b = (np.hamming(K) / K).astype(np.float32)
    try:
        f = G.fir_filter(b, torch.float32, decimate=D)
    except Exception as e:
        print("D=%2d K=%4d: %s" % (D, K, str(e)[:80])); continue
    nn = n - n % D
    y = torch.empty(nn // D, dtype=torch.float32, device="cuda")
    t = steady(lambda: f.process_bulk(x[:nn], y))
    print("D=%2d K=%4d: %7.1f G input samples/s  %6.2f TB/s (4 + 4/D B per input)  %6.1f TFLOP/s useful (2 K / D per input)" % (D, K, nn / t / 1e9, nn * (4 + 4.0 / D) / t / 1e12, 2.0 * K / D * nn / t / 1e12))
