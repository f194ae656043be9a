"""BASELINE configs[2] once per mode, for rocprofv3: gr4hip_fir_iir_process on 2^27 input samples, three calls.  usage: configs2_one.py one|two"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import gnuradio4_amd as G
import gnuradio4_amd.blocks as B
from gnuradio4_amd import capi

mode = capi.FIR_IIR_ONE_LAUNCH if sys.argv[1] == "one" else capi.FIR_IIR_TWO_LAUNCHES
n = (1 << 27) // 7168 * 7168
k = np.arange(1024)
t = np.hamming(1024) * 0.1 * np.sinc(0.1 * (k - 511.5))
x = G.synth_f32(n, seed=42)
b, a = B.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH)
fir, iir = G.fir_filter((t / t.sum()).astype(np.float32), torch.float32, decimate=8), G.iir_filter(b, a)
y = torch.empty(n // 8, dtype=torch.float32, device="cuda")
for _ in range(3):
    B.fir_iir_process(fir, iir, x, y, mode=mode)
torch.cuda.synchronize()
print("ok", float(y[-1]))
