#!/usr/bin/env python
"""developer tool: the headline bench under the strict guard and with the guard off, one line each (for tools/ab_run.sh)"""
import json, subprocess, sys
for g in (0, 2, 0, 2):
    out = subprocess.run([sys.executable, "bench.py", "--no-cpu-baseline", "--no-graph8", "--guard-mode", str(g)], capture_output=True, text=True).stdout
    d = json.loads(out.strip().splitlines()[-1])
    print(f"guard mode {g}: {d['value'] / 1e3:.1f} Gsamples/s  frac {d['roofline']['frac']:.4f}  launch {d['roofline']['avg_launch_ms']:.4f} ms")
