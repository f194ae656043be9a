#!/bin/bash
# developer tool: PMC counters for the fused chain kernel (separate passes; no tracing domains combined with --pmc)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/pmc
mkdir -p $OUT
CMD="python $R/bench.py --steps 1 --warmup 1 --log2-samples 28 --no-cpu-baseline"
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT -o p1 --output-format csv -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU -d $OUT -o p2 --output-format csv -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT -o p3 --output-format csv -- $CMD > $OUT/p3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT -o p4 --output-format csv -- $CMD > $OUT/p4.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT -o p5 --output-format csv -- $CMD > $OUT/p5.log 2>&1
ls -R $OUT | head -30
