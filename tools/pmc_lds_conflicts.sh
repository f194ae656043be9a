#!/bin/bash
# developer tool: LDS bank-conflict share (SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE) of every kernel the given script launches.  usage: pmc_lds_conflicts.sh <tag> <script.py> [args]
TAG=$1; shift
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc_lds_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_BUSY_CU_CYCLES -d $OUT -o p --output-format csv -- python $R/$* > $OUT/log.txt 2>&1
python - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/**/*_counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "gr4::" in r["Kernel_Name"]:
            agg[r["Kernel_Name"][:95]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, m in sorted(agg.items()):
    a = sum(m["SQ_LDS_IDX_ACTIVE"]) / max(len(m["SQ_LDS_IDX_ACTIVE"]), 1)
    c = sum(m["SQ_LDS_BANK_CONFLICT"]) / max(len(m["SQ_LDS_BANK_CONFLICT"]), 1)
    b = sum(m["SQ_BUSY_CU_CYCLES"]) / max(len(m["SQ_BUSY_CU_CYCLES"]), 1)
    if a > 0:
        print("%-95s conflicts %5.1f %% of LDS cycles; LDS busy %5.1f %% of CU-busy cycles" % (k, 100 * c / a, 100 * a / max(b, 1)))
PY
