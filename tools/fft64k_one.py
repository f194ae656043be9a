#!/usr/bin/env python
"""developer tool: 65536-point transforms for rocprofv3 (2^27 points, ten launches of spectrum and mag2)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnuradio4_amd as G
N = 65536; frames = (1 << 27) // N
x = G.synth_c32(frames * N, seed=5)
F = G.FFT(N, "None")
o1 = torch.empty((frames, N), dtype=torch.complex64, device="cuda"); o2 = torch.empty((frames, N), dtype=torch.float32, device="cuda")
for _ in range(10):
    F.spectrum(x, o1); F.mag2(x, o2)
torch.cuda.synchronize()
