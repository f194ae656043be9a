#!/bin/bash
# Collect the rocprofv3 evidence for a round: kernel-trace stats of the default bench command, then PMC passes (separate runs,
# counters only) on a shorter stream.  Writes under gpurun_out/prof_$1/ ; summaries are copied to profiles/ by tools/summarize_profiles.py
TAG=${1:-r01}
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- python $R/bench.py --no-cpu-baseline --no-live-traffic --no-secondary > $OUT/bench_under_trace.log 2>&1
CMD="python $R/bench.py --steps 1 --warmup 1 --log2-samples 28 --no-cpu-baseline --no-live-traffic --no-secondary --no-graph8 --no-hann-row"  # (no Hann / guard rows: their 2^27-sample launches of the same kernel would be averaged in)
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU -d $OUT -o pmc_sq1 --output-format csv -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU -d $OUT -o pmc_sq2 --output-format csv -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT -o pmc_fetch --output-format csv -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT -o pmc_write --output-format csv -- $CMD > $OUT/pmc4.log 2>&1
rocprofv3 --pmc SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_VALU_MFMA_MOPS_F32 -d $OUT -o pmc_mfma --output-format csv -- $CMD > $OUT/pmc6.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE -d $OUT -o pmc_grbm --output-format csv -- $CMD > $OUT/pmc5.log 2>&1
cd $R && python bench.py > $OUT/bench.json 2> $OUT/bench.err
ls $OUT
