#!/usr/bin/env python
"""developer tool: the fused time-domain chain (GR4HIP_CHAIN_FUSED_TD) beside the fused fast convolution (FUSED_FD) and the FIR kernel + FFT kernel pair"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from _timing import steady
import gnuradio4_amd as G
from gnuradio4_amd import capi
n = 1 << 27
x = G.synth_c32(n, seed=5)
for N, ntaps, window in ((1024, 64, "Hann"), (1024, 64, "None"), (1024, 32, "Hann"), (256, 64, "Hann"), (4096, 64, "Hann"), (2048, 128, "Hann"), (1024, 256, "Hann"), (4096, 256, "None")):
    kk = np.arange(ntaps); t = np.hamming(ntaps) * 0.2 * np.sinc(0.2 * (kk - (ntaps - 1) / 2)); t = (t / t.sum()).astype(np.float32)
    out = torch.empty((n // N, N), dtype=torch.float32, device="cuda")
    r = []
    for algo in (capi.CHAIN_FUSED_TD, capi.CHAIN_FUSED_FD, capi.CHAIN_TIME_DOMAIN):
        ch = G.Chain(t, N, window, algo)
        r.append(n / steady(lambda: ch.process_bulk(x, out)) / 1e9)
    print("%4d-pt %-5s %3d taps: fused time domain %6.1f | fused fast convolution %6.1f | FIR kernel + FFT kernel %6.1f Gsamples/s  (fused TD: %.2f TB/s at 12 B/sample)" % (N, window, ntaps, r[0], r[1], r[2], r[0] * 12 / 1e3))
