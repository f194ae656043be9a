"""developer tool: fir_filter<complex<float>> beyond 256 taps (slices of 256 on the two-term f16 kernel since round 5; GR4HIP_FIR_NO_F16X2=1: the f32 matrix-pipe kernel they took before), 2^26 samples, steady state"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import gnuradio4_amd as G
from gnuradio4_amd import capi
from _timing import steady
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
x = G.synth_c32(1 << 26, seed=1); y = torch.empty_like(x)
for nt in (256, 257, 384, 512, 768, 1024, 1280, 1792, 1800):
    row = []
    for sw in (0, 1):
        capi.developer_switch("GR4HIP_FIR_NO_F16X2", sw)
        f = G.fir_filter(lowpass(nt, 0.05), torch.complex64)
        row.append(x.numel() / steady(lambda: f.process_bulk(x, y)) / 1e9)
    capi.developer_switch("GR4HIP_FIR_NO_F16X2", 0)
    print(f"{nt:5d} taps: {row[0]:6.1f} Gsamples/s   (without the f16 kernels: {row[1]:6.1f})", flush=True)
