"""developer tool: the fused time-domain chain (AUTO, <= 64 taps) on narrow filters -- does its per-segment verdict mark white noise?  64 taps -> 1024-point Hann |X|^2, 2^27 samples"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import gnuradio4_amd as G
from _timing import steady
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
n, N = 1 << 27, 1024
x = G.synth_c32(n, seed=3); m = torch.empty((n // N, N), dtype=torch.float32, device="cuda")
out = []
for fc in (0.1, 0.03, 0.01, 0.003):
    b = lowpass(64, fc)
    ch = G.Chain(b, N, "Hann")
    out.append(f"fc {fc} (sum b^2 {float(np.sum(b.astype(np.float64) ** 2)):.4f}): {n / steady(lambda: ch.process_bulk(x, m)) / 1e9:.0f}")
print("  ".join(out), flush=True)
