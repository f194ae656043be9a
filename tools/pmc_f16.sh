#!/bin/bash
# developer tool: counters of the two-term f16 FIR kernel on BASELINE configs[3] (counter-only passes + one kernel trace).  usage: pmc_f16.sh
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_f16
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/fir_batched_prof.py"
timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT -o trace -- $CMD > $OUT/trace.log 2>&1
timeout 200 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT -o p1 --output-format csv -- $CMD > $OUT/p1.log 2>&1
timeout 200 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAIT_INST_ANY SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS -d $OUT -o p2 --output-format csv -- $CMD > $OUT/p2.log 2>&1
timeout 200 rocprofv3 --pmc FETCH_SIZE -d $OUT -o p3 --output-format csv -- $CMD > $OUT/p3.log 2>&1
timeout 200 rocprofv3 --pmc WRITE_SIZE -d $OUT -o p4 --output-format csv -- $CMD > $OUT/p4.log 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "f16x2" in r["Kernel_Name"]:
            agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
for c, v in sorted(agg.items()):
    print(f"{c:28s} n={len(v)} mean={sum(v)/len(v):.6g}")
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gr4::" in r["Name"]:
            print("trace:", r["Name"][:80], "calls", r["Calls"], "avg_us", float(r["AverageNs"]) / 1e3)
PY
