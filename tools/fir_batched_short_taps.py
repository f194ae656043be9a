#!/usr/bin/env python
"""developer tool: 64-channel batched FIR at 33..64 taps, bf16 three-term kernel (default) or the f32 MFMA kernel (GR4HIP_FIR_BATCHED_BF16_MIN_TAPS=65)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from _timing import steady
import gnuradio4_amd as G
nch, n = 64, 1 << 21
x = torch.randn((nch, n), dtype=torch.float32, device="cuda"); y = torch.empty_like(x)
for ntaps in (33, 48, 64, 65):
    b = (np.random.default_rng(ntaps).standard_normal((nch, ntaps)) / np.sqrt(ntaps)).astype(np.float32)
    f = G.FirBatched(b)
    t = steady(lambda: f.process_bulk(x, y))
    print("%3d taps x %d channels: %6.1f Gsamples/s" % (ntaps, nch, nch * n / t / 1e9))
