#!/usr/bin/env python
"""developer tool: IIR cascade rates -- segment-sequential kernel (default for long spans) vs the decoupled look-back kernel (GR4HIP_IIR_LOOKBACK=1)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _timing import steady
import gnuradio4_amd as G
from gnuradio4_amd import capi
def rate(f, x, y):
    return x.numel() / steady(lambda: f.process_bulk(x, y)) / 1e9  # back to back at settled clocks (tools/_timing.py)
for log2n in [int(v) for v in os.environ.get('IIR_LOG2N', '24,26,27').split(',')]:
    n = 1 << log2n
    x = G.synth_f32(n, seed=1); y = torch.empty_like(x)
    b4, a4 = G.blocks.design_iir(capi.LOWPASS, 8, 0.05, float("nan"), 1.0, capi.BUTTERWORTH)
    for name, mk in (("4 biquads (Butterworth-8, fc 0.05)", lambda: G.iir_filter(b4, a4)), ("1-pole low-pass a = 0.95", lambda: G.iir_filter([[0.05]], [[1.0, -0.95]]))):
        out = []
        for env in (None, "1"):
            capi.developer_switch("GR4HIP_IIR_LOOKBACK", 1 if env else 0)
            out.append(rate(mk(), x, y))
        capi.developer_switch("GR4HIP_IIR_LOOKBACK", 0)
        print("2^%d %-38s sequential runs %7.1f Gsamples/s (%.2f TB/s) | look-back %7.1f Gsamples/s" % (log2n, name, out[0], out[0] * 8 / 1e3, out[1]))
