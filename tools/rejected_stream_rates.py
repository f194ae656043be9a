#!/usr/bin/env python
"""developer tool (round 5): what a stream costs on which EVERY segment is marked (a tone 60 dB above the noise, far outside the pass band) -- the main kernel + the
float64 second evaluation (fir_exact_kernel) -- beside an ordinary stream through the same filter.  Gsamples/s (input rate), back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnuradio4_amd as G

def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)

def rate(fn, n, reps=5):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(3):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); b.synchronize()
        ms.append(a.elapsed_time(b) / reps)
    return n / (sorted(ms)[1] * 1e-3) / 1e9

n = 1 << 26
for cplx, D, taps in ((False, 1, 64), (False, 1, 256), (False, 1, 1024), (True, 1, 256), (False, 8, 1024), (False, 16, 256), (True, 8, 128), (False, 4, 128)):
    dt = torch.complex64 if cplx else torch.float32
    base = G.synth_c32(n, seed=3) if cplx else G.synth_f32(n, seed=3)
    k = torch.arange(n, device="cuda", dtype=torch.float32)
    ph = 2 * np.pi * ((0.31 * k) % 1.0)
    tone = (torch.polar(torch.full_like(ph, 1000.0), ph) if cplx else 1000.0 * torch.cos(ph))
    loud = (base * 0.05 + tone).to(dt)
    quiet = base
    b = lowpass(taps, 0.4 / D if D > 1 else 0.05)
    y = torch.empty(n // D, dtype=dt, device="cuda")
    f1, f2 = G.fir_filter(b, dt, decimate=D), G.fir_filter(b, dt, decimate=D)
    print(f"{'complex' if cplx else 'float  '} D={D:2d} taps={taps:4d}: ordinary stream {rate(lambda: f1.process_bulk(quiet, y), n):7.1f}   every segment marked {rate(lambda: f2.process_bulk(loud, y), n):7.1f} Gsamples/s", flush=True)
    del base, loud, tone, k, ph
