#!/usr/bin/env python
"""developer tool: microseconds per call on small spans (what a scheduler's 4 Ki .. 64 Ki-sample work() chunks cost: launch count, not arithmetic)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import steady
import gnuradio4_amd as G
from gnuradio4_amd import capi
taps = (np.hamming(64) / 34).astype(np.float32)
b, a = G.blocks.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH)
for n in (4096, 65536, 1 << 18):
    xf, xc = G.synth_f32(n), G.synth_c32(n)
    yf, yc = torch.empty_like(xf), torch.empty_like(xc)
    m = torch.empty(n, dtype=torch.float32, device="cuda")
    cases = [("fir_filter<float> 64 taps", G.fir_filter(taps, torch.float32), xf, yf), ("fir_filter<complex> 64 taps", G.fir_filter(taps, torch.complex64), xc, yc),
             ("iir 4 biquads", G.iir_filter(b, a), xf, yf), ("rotator", None, None, None), ("MultiplyConst<float>", None, None, None), ("chain 64 taps -> 1024 Hann", None, None, None),
             ("FFT 1024 Hann mag2", None, None, None)]
    rot, ch, F = G.Rotator(phase_increment=0.1), G.Chain(taps, 1024, "Hann"), G.FFT(1024, "Hann")
    out = []
    for name, blk, x, y in cases:
        if name == "rotator": fn = lambda: rot.process_bulk(xc)
        elif name.startswith("Multiply"): fn = lambda: G.math_const("Multiply", xf, 2.0)
        elif name.startswith("chain"): fn = lambda: ch.process_bulk(xc, m.view(n // 1024, 1024))
        elif name.startswith("FFT"): fn = lambda: F.mag2(xc, m.view(n // 1024, 1024))
        else: fn = (lambda blk=blk, x=x, y=y: blk.process_bulk(x, y))
        out.append("%s %.1f" % (name, steady(fn, warm_s=0.03, time_s=0.06) * 1e6))
    print("n = %6d, us per call: " % n + " | ".join(out))
