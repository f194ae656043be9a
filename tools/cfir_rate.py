#!/usr/bin/env python
"""developer tool: fir_filter<complex<float>> (fast convolution, chain_fd_kernel<kModeFir>) beside the headline chain at the same sizes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import steady
import gnuradio4_amd as G
def timeit(fn):
    return steady(fn)  # back to back at settled clocks (tools/_timing.py)
k = np.arange(256); taps = np.hamming(256) * 0.2 * np.sinc(0.2 * (k - 127.5)); taps = (taps / taps.sum()).astype(np.float32)  # bench.py's taps
for log2n in (28, 27, 28):
    n = 1 << log2n
    x = G.synth_c32(n); y = torch.empty(n, dtype=torch.complex64, device="cuda"); m = torch.empty(n, dtype=torch.float32, device="cuda")
    f = G.fir_filter(taps, torch.complex64)
    t = timeit(lambda: f.process_bulk(x, y))
    c = G.Chain(taps, 8192)
    tc = timeit(lambda: c.process_bulk(x, m))
    print("2^%d: complex FIR %.3f ms = %6.1f Gsamples/s (%.2f TB/s at 16 B) | chain -> mag2 %.3f ms = %6.1f Gsamples/s (%.2f TB/s at 12 B)" % (log2n, t * 1e3, n / t / 1e9, n * 16 / t / 1e12, tc * 1e3, n / tc / 1e9, n * 12 / tc / 1e12))
    del x, y, m
from gnuradio4_amd import capi
for ntaps in (256, 64):
    n = 1 << 27
    kk = np.arange(ntaps); t = np.hamming(ntaps) * 0.2 * np.sinc(0.2 * (kk - (ntaps - 1) / 2)); t = (t / t.sum()).astype(np.float32)
    x = G.synth_c32(n); y = torch.empty(n, dtype=torch.complex64, device="cuda"); m = torch.empty(n, dtype=torch.float32, device="cuda")
    f = G.fir_filter(t, torch.complex64); f.set_algo(capi.FIR_TIME_DOMAIN)
    tt = timeit(lambda: f.process_bulk(x, y))
    c = G.Chain(t, 8192, "None", capi.CHAIN_TIME_DOMAIN)
    tc = timeit(lambda: c.process_bulk(x, m))
    print("time domain, %3d taps: complex FIR %.3f ms = %6.1f Gsamples/s (%.1f TFLOP/s) | chain (FIR kernel + FFT kernel) %.3f ms = %6.1f Gsamples/s" % (ntaps, tt * 1e3, n / tt / 1e9, n * 4 * ntaps / tt / 1e12, tc * 1e3, n / tc / 1e9))
    del x, y, m
