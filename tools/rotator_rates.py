#!/usr/bin/env python
"""developer tool: Rotator throughput vs phase increment (leaping walker below 0.25 rad, plain walker above; GR4HIP_ROTATOR_WALK=1 forces the plain one)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnuradio4_amd as G
n = 1 << 24
x = G.synth_c32(n)
for inc in (1e-4, 1e-3, 0.01, 0.05, 0.1, 0.2, 0.24, 0.3, 1.0):
    r = G.Rotator(phase_increment=inc)
    r.process_bulk(x[: 1 << 20])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); r.process_bulk(x); b.record(); b.synchronize()
    print("inc %-8g %9.1f Msamples/s" % (inc, n / a.elapsed_time(b) / 1e3))
