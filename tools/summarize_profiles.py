#!/usr/bin/env python
"""Turn gpurun_out/prof_<tag>/ (rocprofv3 outputs) into the committed text summaries under profiles/.
usage: summarize_profiles.py <tag> <kernel-name-for-the-file> [<kernel-substring to match, default: the name>]"""
import collections
import csv
import glob
import os
import sqlite3
import sys

tag, kern = sys.argv[1], sys.argv[2]
match = sys.argv[3] if len(sys.argv) > 3 else kern  # (bench.py also launches the Hann instantiation: chain_fd_kernel<1, 13>)
src = f"gpurun_out/prof_{tag}"
os.makedirs("profiles", exist_ok=True)
lines = []
db = sqlite3.connect(os.path.join(src, "trace_results.db"))
rows = db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall()
lines.append(f"# rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline   (round tag {tag}; durations in microseconds)")
lines.append("# median_us is the figure to read: avg_us mixes dispatches of different launch sizes (the guard row's short launches, the 2^28-sample counter passes) into one mean (VERDICT r05)")
lines.append(f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'median_us':>10} {'%':>6}  kernel")
for name, calls, total, avg, pct in rows:
    ds = sorted((e - b) / 1e3 for b, e in db.execute("select start, end from kernels where name = ?", (name,)).fetchall())
    med = ds[len(ds) // 2] if ds else float("nan")
    lines.append(f"{calls:7d} {total:12.1f} {avg:10.2f} {med:10.2f} {pct:6.2f}  {name[:150]}")
durs = [(e - b) / 1e3 for b, e in db.execute("select start, end from kernels where name like ? order by start", (f"%{match}%",)).fetchall()]
if durs:
    sd = sorted(durs)
    lines.append("")
    lines.append(f"# per-dispatch durations of '{match}' in launch order (us): the clocks ramp over the first ~20 launches after the idle gap, the")
    lines.append(f"# steady-state tail is what bench.py's HIP events report for the last timed step (roofline.avg_launch_ms)")
    lines.append("#   " + " ".join(f"{d:.0f}" for d in durs))
    lines.append(f"#   min {sd[0]:.1f}  median {sd[len(sd)//2]:.1f}  max {sd[-1]:.1f}  mean of last 4 {sum(durs[-4:])/4:.1f}")
lines.append("")
lines.append(f"# rocprofv3 --pmc <counters> (separate passes, counters only) -- python bench.py --steps 1 --warmup 1 --log2-samples 28 --no-cpu-baseline")
lines.append(f"# per-dispatch means for kernels matching '{match}' (one dispatch = one launch of 2^28 samples)")
for f in sorted(glob.glob(os.path.join(src, "pmc_*_counter_collection.csv"))):
    agg = collections.defaultdict(list)
    rows_k = [r for r in csv.DictReader(open(f)) if match in r["Kernel_Name"]]
    full = max((int(r["Grid_Size"]) for r in rows_k), default=0)  # the guard's probe launch (first call of a stream: 8 frames) is a dispatch too: full-size launches only
    for r in rows_k:
        if int(r["Grid_Size"]) == full:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, v in agg.items():
        lines.append(f"{k:24s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
fetch = [l for l in lines if l.startswith("FETCH_SIZE")]
write = [l for l in lines if l.startswith("WRITE_SIZE")]
if fetch and write:
    fk = float(fetch[0].split("mean=")[1]); wk = float(write[0].split("mean=")[1])
    lines.append("")
    lines.append(f"# HBM traffic per launch (FETCH_SIZE / WRITE_SIZE are in KiB): read {fk*1024/1e6:.1f} MB, write {wk*1024/1e6:.1f} MB")
    lines.append("# (MI355X_MICROARCH.md: FETCH_SIZE under-reports 16-B/lane streaming reads by 2x on gfx950; this kernel's reads are 16-B/lane LDS-DMA)")
    lines.append(f"# corrected: {2*fk*1024/1e6:.1f} MB read + {wk*1024/1e6:.1f} MB written = {(2*fk+wk)*1024/1e6:.1f} MB per launch of 2^28 samples (algorithmic: 3221.2 MB)")
    if kern == "chain_fd_kernel":
        import json
        json.dump({"kernel": "gr4::chain_fd_kernel<0, 13>", "samples_per_launch": 1 << 28, "hbm_bytes_per_launch": int((2 * fk + wk) * 1024),
                   "how": f"rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate counter-only passes (profiles/{tag}_{kern}_rocprof_summary.txt): FETCH_SIZE {fk:.0f} KiB x2 "
                          "(gfx950 under-reports 16-B/lane streaming reads by 2x, MI355X_MICROARCH.md HBM section) + WRITE_SIZE " f"{wk:.0f} KiB",
                   "algorithmic_bytes_per_launch": 12 << 28}, open("profiles/latest_pmc.json", "w"), indent=1)
bj = os.path.join(src, "bench.json")
if os.path.exists(bj):
    lines.append("")
    lines.append("# python bench.py (un-profiled, same box)")
    lines.append(open(bj).read().strip().splitlines()[-1])
out = f"profiles/{tag}_{kern}_rocprof_summary.txt"
open(out, "w").write("\n".join(lines) + "\n")
print(open(out).read())
