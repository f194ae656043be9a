"""developer tool: the chain's kernel pair (GR4HIP_CHAIN_UNFUSED, 256 taps / 8192-point frames) a few times at two cut-offs, for a rocprofv3 kernel trace (which kernels, how long each)"""
import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np, torch
import gnuradio4_amd as G
from gnuradio4_amd import capi
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
n, N = 1 << 27, 8192
x = G.synth_c32(n, seed=3)
m = torch.empty((n // N, N), dtype=torch.float32, device="cuda")
for fc in (0.05, 0.005):
    ch = G.Chain(lowpass(256, fc), N, "None", capi.CHAIN_UNFUSED)
    for _ in range(6): ch.process_bulk(x, m)
    torch.cuda.synchronize()
