#!/usr/bin/env python
"""developer tool: rates across parameter regimes that the headline numbers do not cover (IIR section counts, FFT sizes off the fast path)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch, scipy.signal as sps
from _timing import steady
import gnuradio4_amd as G
n = 1 << 26
x = G.synth_f32(n); y = torch.empty_like(x)
for order in (2, 4, 6, 8, 10, 12, 16):
    sos = sps.butter(order, 0.1, output="sos")
    try:
        f = G.iir_filter(sos[:, :3], sos[:, 3:])
        t = steady(lambda: f.process_bulk(x, y))
        print("iir %2d biquads: %7.1f Gsamples/s (%.2f TB/s)" % (len(sos), n / t / 1e9, 8.0 * n / t / 1e12))
    except Exception as e:
        print("iir %2d biquads: %s" % (len(sos), str(e)[:90]))
for order in (3, 4, 8):
    b, a = sps.butter(order, 0.2)
    try:
        f = G.iir_filter(b[None, :], a[None, :])
        t = steady(lambda: f.process_bulk(x, y))
        print("iir one section of order %d: %7.1f Gsamples/s" % (order, n / t / 1e9))
    except Exception as e:
        print("iir one section of order %d: %s" % (order, str(e)[:90]))
xc = G.synth_c32(1 << 24)
for N in (1000, 1536, 3000, 10000, 16384, 65536, 100000, 1 << 20):
    frames = xc.numel() // N
    F = G.FFT(N, "Hann")
    out = torch.empty((frames, N), dtype=torch.float32, device="cuda")
    t = steady(lambda: F.mag2(xc[: frames * N], out))
    print("FFT %7d -> mag2: %7.1f Gsamples/s (%.2f TB/s at 12 B)" % (N, frames * N / t / 1e9, 12.0 * frames * N / t / 1e12))
