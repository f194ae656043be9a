#!/usr/bin/env python
"""gpurun_out/prof_secondary_stats (tools/profile_secondary_stats.sh) -> profiles/r01_secondary_configs.json + r01_secondary_kernel_stats.txt"""
import glob, os, shutil, sqlite3, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_secondary_stats"
shutil.copy(os.path.join(src, "configs.json"), "profiles/r01_secondary_configs.json")
db = sqlite3.connect(glob.glob(os.path.join(src, "**", "*.db"), recursive=True)[0])
out = ["# rocprofv3 --kernel-trace --stats -- python tools/bench_configs.py  (secondary BASELINE configs + stand-alone blocks; the wall-clock rates are in r01_secondary_configs.json; torch helper kernels omitted)",
       "# rocprofv3 --kernel-trace --stats summary (top_kernels of the results database); durations in microseconds",
       f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'%':>6}  kernel"]
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall():
    if "gr4::" not in name:
        continue
    out.append(f"{calls:7d} {total:12.1f} {avg:10.2f} {pct:6.2f}  {name[:170]}")
open("profiles/r01_secondary_kernel_stats.txt", "w").write("\n".join(out) + "\n")
print("profiles/r01_secondary_kernel_stats.txt", len(out) - 3, "kernels")
