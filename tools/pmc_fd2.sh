#!/bin/bash
# developer tool: second-level PMC counters for the fused chain kernel (instruction fetch, LDS stalls, scalar/misc issue)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc2
mkdir -p $OUT
CMD="python $R/bench.py --steps 1 --warmup 1 --log2-samples 28 --no-cpu-baseline"
rocprofv3 --pmc SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM -d $OUT -o q1 --output-format csv -- $CMD > $OUT/q1.log 2>&1
rocprofv3 --pmc SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_LDS_DATA_FIFO_FULL SQ_LDS_CMD_FIFO_FULL SQ_INST_LEVEL_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM -d $OUT -o q2 --output-format csv -- $CMD > $OUT/q2.log 2>&1
rocprofv3 --pmc SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_INSTS_BRANCH SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_WAIT_ANY -d $OUT -o q3 --output-format csv -- $CMD > $OUT/q3.log 2>&1
python - <<'PY'
import csv, glob, os, collections
out = os.environ.get("GRAFT_REPO_ROOT", ".") + "/gpurun_out/pmc2"
for f in sorted(glob.glob(out + "/**/*counter_collection.csv", recursive=True)):
    acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:40]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    for k, d in acc.items():
        if "chain" in k: print(os.path.basename(f), k, {c: v for c, v in d.items()})
PY
