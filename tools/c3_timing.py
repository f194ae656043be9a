"""developer tool: BASELINE configs[3] (64 channels x 256-tap float FIR x 2^22 samples) and the single-stream 256-tap float / complex FIR, steady state, one line (for A/B loops over builds)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import gnuradio4_amd as G
from _timing import steady

def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
nch, ntaps, n = 64, 256, 1 << 22
xb = torch.stack([G.synth_f32(n, seed=42 + c) for c in range(nch)]); yb = torch.empty_like(xb)
fb = G.FirBatched(np.stack([lowpass(ntaps, 0.05 + 0.005 * c) for c in range(nch)]))
out = [f"configs[3] {nch * n / steady(lambda: fb.process_bulk(xb, yb)) / 1e9:.0f}"]
x = G.synth_f32(1 << 28, seed=1); y = torch.empty_like(x); f = G.fir_filter(lowpass(256, 0.05), torch.float32)
out.append(f"float 256 taps {x.numel() / steady(lambda: f.process_bulk(x, y)) / 1e9:.0f}")
xc = G.synth_c32(1 << 27, seed=1); yc = torch.empty_like(xc); fc = G.fir_filter(lowpass(256, 0.05), torch.complex64); fc.set_algo(G.capi.FIR_TIME_DOMAIN)
out.append(f"complex 256 taps {xc.numel() / steady(lambda: fc.process_bulk(xc, yc)) / 1e9:.0f}")
print("  ".join(out), flush=True)
