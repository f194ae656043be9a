#!/usr/bin/env python
"""developer tool: rates of the float64 instantiations (csrc/f64.hip: plain FP64 kernels, not the tuned float32 paths)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import steady
import gnuradio4_amd as G
n = 1 << 24
x = torch.randn(n, dtype=torch.float64, device="cuda"); y = torch.empty_like(x)
for K in (32, 64, 128, 256, 1024):
    f = G.fir_filter(np.hanning(K) / K, torch.float64)
    t = steady(lambda: f.process_bulk(x, y))
    print("fir_filter<double> %4d taps: %7.2f Gsamples/s  %.2f TFLOP/s (FP64)" % (K, n / t / 1e9, 2.0 * K * n / t / 1e12))
f = G.fir_filter(np.hanning(1024) / 1024, torch.float64, decimate=8); yd = torch.empty(n // 8, dtype=torch.float64, device="cuda")
t = steady(lambda: f.process_bulk(x, yd))
print("fir_filter<double> 1024 taps decim 8: %7.2f G input samples/s" % (n / t / 1e9))
import scipy.signal as sps
sos = sps.butter(8, 0.1, output="sos")
for name, (b, a) in (("4 biquads", (sos[:, :3], sos[:, 3:])), ("1 pole", (np.array([[0.05, 0.0]]), np.array([[1.0, -0.95]])))):
    f = G.iir_filter(b, a, dtype=torch.float64)
    t = steady(lambda: f.process_bulk(x, y))
    print("iir_filter<double> %-10s: %7.2f Gsamples/s (%.2f TB/s at 16 B/sample)" % (name, n / t / 1e9, 16.0 * n / t / 1e12))
for N in (1024, 8192):
    F = G.FFT(N, "Hann", dtype=torch.float64)
    t = steady(lambda: F.process_bulk(x, ranges=False))
    print("FFT<double> %5d Hann -> DataSet signals: %7.2f G real samples/s" % (N, n / t / 1e9))
xc = torch.randn(n, dtype=torch.complex128, device="cuda")
r = G.Rotator(phase_increment=0.37, dtype=torch.complex128)
t = steady(lambda: r.process_bulk(xc))
print("Rotator<complex<double>>: %7.2f Gsamples/s (%.2f TB/s at 32 B/sample)" % (n / t / 1e9, 32.0 * n / t / 1e12))
