#!/usr/bin/env python
"""developer tool: what the dynamic-range guard costs the decimate-by-8 frequency-domain FIR per call (strict / deferred / off), 2^27 input samples"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _timing import steady
import numpy as np, torch
import gnuradio4_amd as G
from gnuradio4_amd import capi
n = 1 << 27
x = G.synth_f32(n, seed=42)
k = np.arange(1024, dtype=np.float64); w = np.hamming(1024)
t = w * 0.1 * np.sinc(0.1 * (k - 511.5)); b = (t / t.sum()).astype(np.float32)
yd = torch.empty(n // 8, dtype=torch.float32, device="cuda")
for name, mode in (("strict", capi.GUARD_STRICT), ("deferred", capi.GUARD_DEFERRED), ("off", capi.GUARD_OFF)):
    f = G.fir_filter(b, torch.float32, decimate=8)
    capi.check(capi.lib().gr4hip_fir_set_guard_mode(f._h, mode), "mode")
    dt = steady(lambda: f.process_bulk(x, yd))
    print(f"{name:9s} {dt*1e6:7.1f} us per call  {n/dt/1e9:6.1f} G input samples/s")
