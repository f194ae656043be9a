#!/usr/bin/env python
"""developer tool: a few launches of the fused time-domain chain (64 taps -> 1024-pt Hann by default) for rocprofv3; usage: td_one.py [N] [ntaps] [algo]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnuradio4_amd as G
from gnuradio4_amd import capi
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
ntaps = int(sys.argv[2]) if len(sys.argv) > 2 else 64
algo = int(sys.argv[3]) if len(sys.argv) > 3 else capi.CHAIN_FUSED_TD
n = 1 << 27
x = G.synth_c32(n, seed=5)
kk = np.arange(ntaps); t = np.hamming(ntaps) * 0.2 * np.sinc(0.2 * (kk - (ntaps - 1) / 2)); t = (t / t.sum()).astype(np.float32)
out = torch.empty((n // N, N), dtype=torch.float32, device="cuda")
ch = G.Chain(t, N, "Hann", algo)
for _ in range(6):
    ch.process_bulk(x, out)
torch.cuda.synchronize()
