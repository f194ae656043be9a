#!/usr/bin/env python
"""developer tool: single-stream fir_filter rates over the span length (launch geometry at mid-size spans)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import steady
import gnuradio4_amd as G
for dt, name in ((torch.float32, "float"), (torch.complex64, "complex")):
    for K in (256, 64):
        taps = (np.hamming(K) * 0.2 * np.sinc(0.2 * (np.arange(K) - (K - 1) / 2))).astype(np.float32)
        row = []
        for log2n in (18, 20, 22, 24, 26):
            n = 1 << log2n
            x = G.synth_f32(n) if dt == torch.float32 else G.synth_c32(n)
            y = torch.empty_like(x)
            f = G.fir_filter(taps, dt)
            t = steady(lambda: f.process_bulk(x, y))
            row.append("2^%d %6.1f" % (log2n, n / t / 1e9))
        print("fir_filter<%s> %3d taps, Gsamples/s: " % (name, K) + " | ".join(row))
