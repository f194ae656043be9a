#!/bin/bash
# secondary BASELINE configs + stand-alone blocks: wall-clock rates (JSON) and the rocprofv3 kernel-trace statistics of the same script
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_secondary_stats
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $R/tools/bench_configs.py --json $OUT/configs.json > $OUT/configs.log 2>&1
rocprofv3 --kernel-trace --stats -d $OUT -o t -- python $R/tools/bench_configs.py > $OUT/trace.log 2>&1
ls $OUT
