#!/usr/bin/env python
"""developer tool: the last N dispatches of a rocprofv3 --kernel-trace run (start offset and duration in us).  usage: trace_tail.py <dir> [N = 12]"""
import csv, glob, sys
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
last = rows[-(int(sys.argv[2]) if len(sys.argv) > 2 else 12):]
t0 = int(last[0]["Start_Timestamp"])
for r in last:
    print(f'{r["Kernel_Name"][:70]:70s} start {(int(r["Start_Timestamp"]) - t0) / 1e3:9.1f} us  dur {(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3:8.1f} us  grid {r.get("Grid_Size", r.get("Grid_Size_X", "?"))}')
