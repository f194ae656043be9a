#!/usr/bin/env python
"""developer tool: run the 4-biquad IIR cascade on 2^24 floats a few times (for rocprofv3 --kernel-trace --stats)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnuradio4_amd as G
from gnuradio4_amd import capi
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 24)
x = G.synth_f32(n)
y = torch.empty_like(x)
b, a = G.blocks.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH)
iir = G.iir_filter(b, a)
for _ in range(5):
    iir.process_bulk(x, y)
torch.cuda.synchronize()
