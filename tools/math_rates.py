#!/usr/bin/env python
"""developer tool: math blocks (MathOpImpl const ops, MathOpMultiPortImpl n-ary) over the element types: Gsamples/s and TB/s"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from _timing import steady
import gnuradio4_amd as G
nbytes = 1 << 29
for dt in (torch.uint8, torch.int16, torch.int32, torch.int64, torch.float32, torch.float64, torch.complex64, torch.complex128):
    es = torch.empty(0, dtype=dt).element_size()
    n = nbytes // es
    if dt.is_complex:
        x = torch.view_as_complex(torch.randn(n, 2, dtype=torch.float32 if dt == torch.complex64 else torch.float64, device="cuda"))
    elif dt.is_floating_point:
        x = torch.randn(n, dtype=dt, device="cuda")
    else:
        x = torch.randint(1, 100, (n,), dtype=dt, device="cuda")
    for op, val in (("Multiply", 3), ("Divide", 3)):
        try:
            t = steady(lambda: G.math_const(op, x, val))
            print("%-10s %-18s const: %7.1f Gsamples/s  %5.2f TB/s" % (op, str(dt), n / t / 1e9, 2.0 * nbytes / t / 1e12))
        except Exception as e:
            print("%-10s %-18s const: %s" % (op, str(dt), str(e)[:70]))
    ins = [x, x, x]
    try:
        t = steady(lambda: G.math_nary("Add", ins))
        print("%-10s %-18s n=3  : %7.1f Gsamples/s  %5.2f TB/s" % ("Add", str(dt), n / t / 1e9, 4.0 * nbytes / t / 1e12))
    except Exception as e:
        print("Add %s n=3: %s" % (dt, str(e)[:70]))
    del x, ins
