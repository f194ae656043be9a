#!/usr/bin/env python
"""developer tool (round 5): decimate-by-4 on the f16 band-form kernel against the bf16 band kernel it replaces (run with GR4HIP_FIR_DECIM_F16_MIN_TAPS_D4=100000 for the bf16 rates),
float and complex, and the hooked decimators (rotator -> decimate-by-8 complex FIR: the channeliser's front end).  G input samples/s, back-to-back launches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnuradio4_amd as G

def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)

def rate(fn, n, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    ms = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        for _ in range(reps): fn()
        b.record(); b.synchronize()
        ms.append(a.elapsed_time(b) / reps)
    return n / (sorted(ms)[2] * 1e-3) / 1e9

print("# tools/decim_d4_rates.py  GR4HIP_FIR_DECIM_F16_MIN_TAPS_D4 =", os.environ.get("GR4HIP_FIR_DECIM_F16_MIN_TAPS_D4", "(default 33)"))
for cplx in (False, True):
    n = (1 << 27) if not cplx else (1 << 26)
    x = G.synth_c32(n, seed=2) if cplx else G.synth_f32(n, seed=2)
    for D, taps in ((4, 33), (4, 64), (4, 128), (4, 256), (4, 512)):
        f = G.fir_filter(lowpass(taps, 0.4 / D), torch.complex64 if cplx else torch.float32, decimate=D)
        y = torch.empty(n // D, dtype=x.dtype, device="cuda")
        print(f"{'complex' if cplx else 'float  '} D={D} taps={taps:4d}: {rate(lambda: f.process_bulk(x, y), n):7.1f} G input samples/s", flush=True)
    del x
# hooked decimators
nc = 1 << 26
xc = G.synth_c32(nc, seed=2)
for D, taps in ((8, 64), (8, 128), (16, 64), (4, 64), (32, 100), (10, 80), (5, 300)):
    b = lowpass(taps, 0.4 / D)
    yd = torch.empty(nc // D, dtype=torch.complex64, device="cuda")
    xs = xc[: nc // D * D]
    fh = G.fir_filter(b, torch.complex64, decimate=D); fh.set_prologue(G.Merged(torch.complex64, [("Rotator", 0.3, 0.25)]))
    fp = G.fir_filter(b, torch.complex64, decimate=D)
    print(f"complex D={D} taps={taps:4d}: plain {rate(lambda: fp.process_bulk(xs, yd), nc):7.1f}   rotator as load program {rate(lambda: fh.process_bulk(xs, yd), nc):7.1f} G input samples/s", flush=True)
