#!/bin/bash
# BASELINE configs[2] through gr4hip_fir_iir_process, one launch against two: the kernel trace (dispatches per call) and the HBM counters (separate passes)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_configs2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for m in one two; do
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT -o trace_$m -- python $R/tools/configs2_one.py $m > $OUT/trace_$m.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT -o fetch_$m --output-format csv -- python $R/tools/configs2_one.py $m > $OUT/fetch_$m.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE -d $OUT -o write_$m --output-format csv -- python $R/tools/configs2_one.py $m > $OUT/write_$m.log 2>&1
done
ls $OUT
