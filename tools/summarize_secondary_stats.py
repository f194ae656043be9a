#!/usr/bin/env python
"""gpurun_out/prof_secondary_stats (tools/profile_secondary_stats.sh) -> profiles/<tag>_secondary_configs.json + <tag>_secondary_kernel_stats.txt   usage: summarize_secondary_stats.py [src] [tag]"""
import glob, os, shutil, sqlite3, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_secondary_stats"
tag = sys.argv[2] if len(sys.argv) > 2 else "r02"
shutil.copy(os.path.join(src, "configs.json"), f"profiles/{tag}_secondary_configs.json")
db = sqlite3.connect(glob.glob(os.path.join(src, "**", "*.db"), recursive=True)[0])
out = [f"# rocprofv3 --kernel-trace --stats -- python tools/bench_configs.py  (secondary BASELINE configs + stand-alone blocks; the wall-clock rates are in {tag}_secondary_configs.json; torch helper kernels omitted)",
       "# rocprofv3 --kernel-trace --stats summary (top_kernels of the results database); durations in microseconds",
       f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'%':>6}  kernel"]
for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall():
    if "gr4::" not in name:
        continue
    out.append(f"{calls:7d} {total:12.1f} {avg:10.2f} {pct:6.2f}  {name[:170]}")
open(f"profiles/{tag}_secondary_kernel_stats.txt", "w").write("\n".join(out) + "\n")
print(f"profiles/{tag}_secondary_kernel_stats.txt", len(out) - 3, "kernels")
