#!/usr/bin/env python
"""Dump the per-kernel summary (the `--stats` view) of a rocprofv3 rocpd sqlite database as text.
usage: rocpd_summary.py results.db > profiles/<name>.txt"""
import sqlite3
import sys

db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
print(f"# rocprofv3 --kernel-trace --stats summary ({sys.argv[1].split('/')[-1]}); durations in microseconds")
print(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'%':>6}  kernel")
for name, calls, total, avg, pct in rows:
    print(f"{calls:7d} {total:12.1f} {avg:10.2f} {pct:6.2f}  {name}")
