"""developer tool: the kernel pair (CHAIN_UNFUSED: FIR with float32 products -> HBM -> FFT kernel) at the shapes CHAIN_AUTO takes it for, and its neighbours: Gsamples/s, steady state"""
import os, sys
sys.path.insert(0, "."); sys.path.insert(0, "tools")
import numpy as np, torch
import gnuradio4_amd as G
from gnuradio4_amd import capi
from _timing import steady
def lowpass(nt, fc=0.05):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
n = 1 << 27
x = G.synth_c32(n, seed=3)
for nt, N, win in ((256, 16384, "None"), (256, 32768, "Hann"), (64, 16384, "None"), (512, 8192, "None"), (256, 8192, "None"), (256, 1000, "Hann"), (64, 1024, "Hann")):
    m = torch.empty((n // N, N), dtype=torch.float32, device="cuda")
    row = [f"{nt} taps, {N}-point {win}:"]
    for name, algo in (("AUTO", capi.CHAIN_AUTO), ("UNFUSED", capi.CHAIN_UNFUSED)):
        ch = G.Chain(lowpass(nt), N, win, algo)
        for _ in range(3): ch.process_bulk(x[: (n // N) * N], m)
        row.append(f"{name} (algo {ch.algo}) {(n // N) * N / steady(lambda: ch.process_bulk(x[: (n // N) * N], m)) / 1e9:.0f}")
    print("  ".join(row), flush=True)
