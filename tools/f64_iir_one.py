#!/usr/bin/env python
"""developer tool: a few calls of iir_filter<double> (4 biquads, 2^24 samples) for rocprofv3 --kernel-trace --stats"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, scipy.signal as sps
import gnuradio4_amd as G
n = 1 << 24
x = torch.randn(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
sos = sps.butter(8, 0.1, output="sos")
f = G.iir_filter(sos[:, :3], sos[:, 3:], dtype=torch.float64)
for _ in range(8):
    f.process_bulk(x, y)
torch.cuda.synchronize()
