#!/usr/bin/env python
"""developer tool: fir_filter<complex<float>> over the tap count -- default dispatch beside GR4HIP_FIR_TIME_DOMAIN (where the fast convolution stops paying)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from _timing import steady
import gnuradio4_amd as G
from gnuradio4_amd import capi
n = 1 << 27
x = G.synth_c32(n); y = torch.empty(n, dtype=torch.complex64, device="cuda")
for ntaps in (8, 16, 24, 32, 33, 48, 64, 65, 96, 128, 129, 192, 256):
    kk = np.arange(ntaps); t = np.hamming(ntaps) * 0.2 * np.sinc(0.2 * (kk - (ntaps - 1) / 2)); t = (t / t.sum()).astype(np.float32)
    f = G.fir_filter(t, torch.complex64)
    ta = steady(lambda: f.process_bulk(x, y))
    g = G.fir_filter(t, torch.complex64); g.set_algo(capi.FIR_TIME_DOMAIN)
    tt = steady(lambda: g.process_bulk(x, y))
    print("%3d taps: default %6.1f Gsamples/s (%.2f TB/s) | time domain %6.1f Gsamples/s (%.2f TB/s, %.1f TFLOP/s useful)" % (ntaps, n / ta / 1e9, n * 16 / ta / 1e12, n / tt / 1e9, n * 16 / tt / 1e12, n * 4 * ntaps / tt / 1e12))
