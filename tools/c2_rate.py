#!/usr/bin/env python
"""developer tool: BASELINE configs[2] (decimate-by-8 1024-tap FIR + 4 biquads) rates, frequency-domain decimator vs the polyphase MFMA kernel"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import steady
import gnuradio4_amd as G
from gnuradio4_amd import capi
def timeit(fn):
    return steady(fn)  # back to back at settled clocks (tools/_timing.py)
n = 7168 * int(os.environ.get("C2_BLOCKS", "18432"))  # whole blocks: ~2^27 samples (C2_BLOCKS=147456: ~2^30)
x = G.synth_f32(n, seed=42)
k = np.arange(1024); w = np.hamming(1024); t = w * 0.1 * np.sinc(0.1 * (k - 511.5)); taps = (t / t.sum()).astype(np.float32)
b, a = G.blocks.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH)
yd = torch.empty(n // 8, dtype=torch.float32, device="cuda"); yo = torch.empty_like(yd)
iir = G.iir_filter(b, a)
t_iir = timeit(lambda: iir.process_bulk(yd, yo))
for tag, env in (("frequency domain", None), ("polyphase MFMA", "1")):
    capi.developer_switch("GR4HIP_FIR_NO_DECIM_FD", 1 if env else 0)
    fir = G.fir_filter(taps, torch.float32, decimate=8)
    t_fir = timeit(lambda: fir.process_bulk(x, yd))
    print("%-18s FIR %.3f ms = %6.1f G input samples/s (%.2f TB/s at 4.5 B/sample) | + IIR %.3f ms -> configs[2] %6.1f G input samples/s" % (tag, t_fir * 1e3, n / t_fir / 1e9, n * 4.5 / t_fir / 1e12, t_iir * 1e3, n / (t_fir + t_iir) / 1e9))
