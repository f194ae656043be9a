"""developer tool: the f16 band-form decimate-by-8 / 1024-tap kernel's rate with the guard off (for the timing-only builds of tools/ab_dh.sh, whose outputs are garbage), and -- on
the default build -- the guarded rate and the 4-biquad cascade alone.  usage: dh_timing.py [full]"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import torch
import gnuradio4_amd as G
from _timing import steady
from gnuradio4_amd import capi

n = 1 << 27
x = G.synth_f32(n, seed=42)
y = torch.empty(n // 8, dtype=torch.float32, device="cuda")
k = np.arange(1024)
t = np.hamming(1024) * 0.1 * np.sinc(0.1 * (k - 511.5))
t = (t / t.sum()).astype(np.float32)
out = []
f = G.fir_filter(t, torch.float32, decimate=8)
f.set_guard_mode(capi.GUARD_OFF)
out.append(f"guard off {n / steady(lambda: f.process_bulk(x, y)) / 1e9:.0f}")
if len(sys.argv) > 1:
    g = G.fir_filter(t, torch.float32, decimate=8)
    out.append(f"guarded {n / steady(lambda: g.process_bulk(x, y)) / 1e9:.0f}")
    b, a = G.blocks.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH)
    iir = G.iir_filter(b, a)
    yo = torch.empty_like(y)
    tt = steady(lambda: iir.process_bulk(y, yo))
    out.append(f"4 biquads on 2^24: {tt * 1e6:.1f} us = {y.numel() * 8 / tt / 1e12:.2f} TB/s")
    def both():
        g.process_bulk(x, y); iir.process_bulk(y, yo)
    out.append(f"configs[2] pair {n / steady(both) / 1e9:.0f}")
print("  ".join(out), flush=True)
