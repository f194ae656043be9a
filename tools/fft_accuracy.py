#!/usr/bin/env python
"""developer tool: spectrum error of the device FFT against numpy's float64 FFT over the size envelope (max over bins and rms, relative to the spectrum's rms)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnuradio4_amd as G
rng = np.random.default_rng(1)
for N in (1024, 8192, 1 << 14, 1 << 16, 1 << 17, 1 << 18, 1 << 20, 1000, 4095, 6000, 30000, 65535, 100003, 273375, (1 << 19) - 1):
    x = (rng.standard_normal(N) + 1j * rng.standard_normal(N)).astype(np.complex64)
    got = G.FFT(N, "None").spectrum(torch.from_numpy(x).cuda()).cpu().numpy()[0]
    truth = np.fft.fft(x.astype(np.complex128))
    rms = np.sqrt(np.mean(np.abs(truth) ** 2))
    err = np.abs(got - truth)
    print("N = %7d  max %.2e  rms %.2e   (of the spectrum's rms)" % (N, err.max() / rms, np.sqrt(np.mean(err ** 2)) / rms))
