#!/usr/bin/env python
"""gpurun_out/prof_mfma_fir (tools/profile_mfma_fir.sh) -> profiles/r01_fir_mfma_kernel_rocprof_summary.txt"""
import collections, csv, glob, os, sqlite3, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_mfma_fir"
dst = sys.argv[2] if len(sys.argv) > 2 else "profiles/r01_fir_mfma_kernel_rocprof_summary.txt"
KERN = "fir_mfma_kernel<68>"
out = ["# rocprofv3 --kernel-trace --stats -- python tools/fir_batched_prof.py   (BASELINE configs[3]: 64 channels x 256 taps x 2^22 samples, f32 MFMA; durations in us)",
       f"{'calls':>7} {'total_us':>12} {'avg_us':>10} {'%':>6}  kernel"]
dbs = glob.glob(os.path.join(src, "**", "*.db"), recursive=True)
avg_us = None
if dbs:
    db = sqlite3.connect(dbs[0])
    for name, calls, total, avg, pct in db.execute("select name, total_calls, total_duration, average, percentage from top_kernels").fetchall():
        out.append(f"{calls:7d} {total:12.1f} {avg:10.2f} {pct:6.2f}  {name[:140]}")  # top_kernels is in microseconds
        if KERN in name:
            avg_us = avg
acc = collections.defaultdict(list)
for path in sorted(glob.glob(os.path.join(src, "**", "*counter_collection.csv"), recursive=True)):
    for row in csv.DictReader(open(path)):
        if KERN in row["Kernel_Name"]:
            acc[row["Counter_Name"]].append(float(row["Counter_Value"]))
m = {k: sum(v) / len(v) for k, v in acc.items()}
out += ["", f"# rocprofv3 --pmc <counters> (separate counter-only passes); per-dispatch means for {KERN}"]
for k in sorted(m):
    out.append(f"{k:<28} n={len(acc[k]):3d} mean={m[k]:.6g}")
out.append("")
if "SQ_VALU_MFMA_BUSY_CYCLES" in m and "SQ_BUSY_CU_CYCLES" in m:
    out.append(f"# MFMA busy: SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x SQ_BUSY_CU_CYCLES) = {100 * m['SQ_VALU_MFMA_BUSY_CYCLES'] / (4 * m['SQ_BUSY_CU_CYCLES']):.1f} % of the matrix-pipe cycles")
    if avg_us:
        out.append(f"#   (SQ_BUSY_CU_CYCLES / 256 CUs / {avg_us:.0f} us = {m['SQ_BUSY_CU_CYCLES'] / 256 / avg_us / 1e3:.2f} GHz sustained under MFMA load; the counter pass itself runs slower than the traced launch)")
if "SQ_INSTS_VALU_MFMA_MOPS_F32" in m:
    out.append(f"# SQ_INSTS_VALU_MFMA_MOPS_F32 = {m['SQ_INSTS_VALU_MFMA_MOPS_F32']:.4g} per launch (x 512 flop per counted op = {m['SQ_INSTS_VALU_MFMA_MOPS_F32'] * 512 / 1e12:.3f} TFLOP executed per launch);"
               f" expected 64 ch x 2^22 samples x 2 x 272 flop = {64 * 2**22 * 544 / 1e12:.3f} TFLOP")
if avg_us:
    out.append(f"# {64 * 2**22 / avg_us / 1e3:.1f} Gsamples/s, {64 * 2**22 * 544 / avg_us / 1e6:.1f} TFLOP/s executed = {100 * 64 * 2**22 * 544 / avg_us / 1e6 / 157.3:.1f} % of the 157.3 TFLOP/s f32 MFMA peak (kernel-trace duration)")
if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
    out.append(f"# HBM traffic per launch: FETCH_SIZE {m['FETCH_SIZE'] * 1024 / 1e6:.1f} MB x 2 (gfx950 under-count, MI355X_MICROARCH.md HBM section) = {m['FETCH_SIZE'] * 2048 / 1e6:.1f} MB")
    out.append(f"#   = the 1073.7 MB input + the 6.25 % halo (Kp = 256 samples per 4096-sample segment) re-read; WRITE_SIZE {m['WRITE_SIZE'] * 1024 / 1e6:.1f} MB; algorithmic 8 B/sample = 2147.5 MB")
if "SQ_LDS_BANK_CONFLICT" in m and "SQ_LDS_IDX_ACTIVE" in m:
    out.append(f"# LDS: SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE = {m['SQ_LDS_BANK_CONFLICT'] / m['SQ_LDS_IDX_ACTIVE']:.3f} (18/16-padded B-operand reads; the 17/16 padding of the first version measured 0.50);")
    if "SQ_BUSY_CU_CYCLES" in m:
        out.append(f"#   SQ_LDS_IDX_ACTIVE / SQ_BUSY_CU_CYCLES = {100 * m['SQ_LDS_IDX_ACTIVE'] / m['SQ_BUSY_CU_CYCLES']:.0f} % -- LDS is not the limiter, the matrix pipe is")
open(dst, "w").write("\n".join(out) + "\n")
print(dst)
