"""developer tool: the register-window kernel's shapes (fir_poly_kernel: <= 32 taps, float and complex; decimate by 2 / 3 at 64 taps), steady state, one line"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import gnuradio4_amd as G
from _timing import steady
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
x = G.synth_f32(1 << 28, seed=1); xc = G.synth_c32(1 << 27, seed=1)
out = []
for nt in (8, 32):
    f = G.fir_filter(lowpass(nt, 0.1), torch.float32); y = torch.empty_like(x)
    out.append(f"float {nt} taps {x.numel() / steady(lambda: f.process_bulk(x, y)) / 1e9:.0f}")
    fc = G.fir_filter(lowpass(nt, 0.1), torch.complex64); yc = torch.empty_like(xc)
    out.append(f"complex {nt} taps {xc.numel() / steady(lambda: fc.process_bulk(xc, yc)) / 1e9:.0f}")
for D in (2, 3):
    f = G.fir_filter(lowpass(24, 0.4 / D), torch.float32, decimate=D); n = (x.numel() // D) * D; y = torch.empty(n // D, dtype=torch.float32, device="cuda")
    out.append(f"float D={D} 24 taps {n / steady(lambda: f.process_bulk(x[:n], y)) / 1e9:.0f}")
print("  ".join(out), flush=True)
