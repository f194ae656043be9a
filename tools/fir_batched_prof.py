#!/usr/bin/env python
"""developer tool: BASELINE configs[3] (64 channels x 256 taps, f32 MFMA) a few times, for rocprofv3 counter / trace passes"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnuradio4_amd as G
nch, ntaps, n = 64, 256, 1 << 22
rng = np.random.default_rng(0)
taps = (rng.standard_normal((nch, ntaps)) / 16).astype(np.float32)
xb = torch.stack([G.synth_f32(n, seed=42 + c) for c in range(nch)])
yb = torch.empty_like(xb)
fb = G.FirBatched(taps)
for _ in range(4):
    fb.process_bulk(xb, yb)
torch.cuda.synchronize()
