#!/usr/bin/env python
"""developer tool: one launch of each secondary kernel on 2^26-sample inputs, for rocprofv3 counter passes (FETCH_SIZE / WRITE_SIZE ...)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gnuradio4_amd as G
from gnuradio4_amd import capi
n = 1 << 26
xc = G.synth_c32(n)
xr = G.synth_f32(n)
m2 = torch.empty(n, dtype=torch.float32, device="cuda")
for N in (1024, 8192):
    F = G.FFT(N, "Hann")
    for _ in range(2):
        F.mag2(xc, m2.view(n // N, N))
b, a = G.blocks.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH)
iir = G.iir_filter(b, a)
y = torch.empty_like(xr)
for _ in range(2):
    iir.process_bulk(xr, y)
k = np.arange(1024) - 511.5
taps = (np.sinc(0.1 * k) * np.hamming(1024)).astype(np.float32)
fd = G.fir_filter(taps, torch.float32, decimate=8)
yd = torch.empty(n // 8, dtype=torch.float32, device="cuda")
for _ in range(2):
    fd.process_bulk(xr, yd)
fc = G.fir_filter(taps[:256], torch.complex64)
yc = torch.empty_like(xc)
for _ in range(2):
    fc.process_bulk(xc, yc)
ch = G.Chain(taps[:256], 8192, "Hann")
for _ in range(2):
    ch.process_bulk(xc, m2.view(n // 8192, 8192))
# round 2: interpolating FIR, closed-form rotator, the complex direct form on the matrix pipe, float64 FIR
it = G.fir_interpolator(taps[:256], 8, torch.float32)
yi = torch.empty(n, dtype=torch.float32, device="cuda")
for _ in range(2):
    it.process_bulk(xr[: n // 8], yi)
rot = G.Rotator(phase_increment=0.37)
for _ in range(2):
    rot.process_bulk(xc)
ft = G.fir_filter(taps[:256], torch.complex64)
ft.set_algo(capi.FIR_TIME_DOMAIN)
for _ in range(2):
    ft.process_bulk(xc, yc)
x64 = xr[: n // 4].double()
f64 = G.fir_filter(taps[:256].astype(np.float64), torch.float64)
for _ in range(2):
    f64.process_bulk(x64)
torch.cuda.synchronize()
