"""developer tool: fir_filter<float> around and beyond 256 taps (one pass up to 256; slices of 256 on the f16 kernel for longer filters), 2^27 samples, steady state"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
import gnuradio4_amd as G
from _timing import steady
def lowpass(nt, fc):
    k = np.arange(nt); t = np.hamming(nt) * 2 * fc * np.sinc(2 * fc * (k - (nt - 1) / 2)); return (t / t.sum()).astype(np.float32)
if len(sys.argv) > 1: G.capi.developer_switch("GR4HIP_FIR_NO_F16X2", 1)  # (any argument: without the f16 kernels -- what long filters took before they became slices)
x = G.synth_f32(1 << 27, seed=1); y = torch.empty_like(x)
print("  ".join(f"{nt}: {x.numel() / steady(lambda: f.process_bulk(x, y)) / 1e9:.0f}" for nt in (256, 257, 300, 350, 383, 384, 512, 768, 1024, 1100, 2048, 3000, 3840, 3900) for f in [G.fir_filter(lowpass(nt, 0.05), torch.float32)]), flush=True)
