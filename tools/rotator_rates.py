#!/usr/bin/env python
"""developer tool: Rotator throughput: closed-form phase (default) on 2^27 samples, and the bit-exact float recurrence vs phase increment"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gnuradio4_amd as G
n = 1 << 27
x = G.synth_c32(n)
out = torch.empty_like(x)
for inc in (1e-3, 0.37, 7.0):
    r = G.Rotator(phase_increment=inc)
    for _ in range(3):
        r.process_bulk(x)
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(10):
        r.process_bulk(x)
    b.record(); b.synchronize()
    ms = a.elapsed_time(b) / 10
    print("closed_form inc %-8g %9.1f Gsamples/s  %.2f TB/s (16 B/sample)" % (inc, n / ms / 1e6, n * 16 / ms / 1e9))
n = 1 << 24
for inc in (1e-4, 1e-3, 0.01, 0.1, 1.0):
    r = G.Rotator(phase_increment=inc, algo="recurrence")
    r.process_bulk(x[: 1 << 20])
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); r.process_bulk(x[:n]); b.record(); b.synchronize()
    print("recurrence  inc %-8g %9.1f Msamples/s" % (inc, n / a.elapsed_time(b) / 1e3))
