#!/usr/bin/env python
"""developer tool: complex direct-form FIR at 33..64 taps on the f32 MFMA (default) or the bf16 three-term kernel (GR4HIP_CFIR_BF16_MIN_TAPS=33 in the environment)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from _timing import steady
import gnuradio4_amd as G
from gnuradio4_amd import capi
n = 1 << 27
x = G.synth_c32(n); y = torch.empty(n, dtype=torch.complex64, device="cuda")
for ntaps in (8, 12, 16, 20, 24, 28, 32, 33, 48, 64):
    kk = np.arange(ntaps); t = np.hamming(ntaps) * 0.2 * np.sinc(0.2 * (kk - (ntaps - 1) / 2)); t = (t / t.sum()).astype(np.float32)
    g = G.fir_filter(t, torch.complex64); g.set_algo(capi.FIR_TIME_DOMAIN)
    tt = steady(lambda: g.process_bulk(x, y))
    print("%3d taps: %6.1f Gsamples/s" % (ntaps, n / tt / 1e9))
