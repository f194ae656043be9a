#!/bin/bash
# developer tool: SQ / LDS / MFMA counters of the fused time-domain chain (counter-only passes).  usage: pmc_td.sh <tag> [td_one.py args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmctd_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/td_one.py $*"
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT -o p1 --output-format csv -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU -d $OUT -o p2 --output-format csv -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA -d $OUT -o p3 --output-format csv -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_MEM_VIOLATIONS SQ_WAVE_DEP_WAIT SQ_INST_LEVEL_LDS SQ_LEVEL_WAVES -d $OUT -o p4 --output-format csv -- $CMD > $OUT/p4.log 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(list)
for f in glob.glob("$OUT/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "chain" in r["Kernel_Name"]:
            agg[(r["Kernel_Name"][:40], r["Counter_Name"])].append(float(r["Counter_Value"]))
print("== $TAG $*")
for (k, c), v in sorted(agg.items()):
    print(f"{k:40s} {c:28s} n={len(v)} mean={sum(v)/len(v):.5g}")
PY
grep -i "chain" $OUT/*kernel_stats.csv | head -3
