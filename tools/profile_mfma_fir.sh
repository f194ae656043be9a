#!/bin/bash
# rocprofv3 evidence for the batched MFMA FIR (BASELINE configs[3]): kernel-trace stats, then counter-only passes
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_mfma_fir
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/fir_batched_prof.py"
rocprofv3 --kernel-trace --stats -d $OUT -o trace -- $CMD > $OUT/trace.log 2>&1
rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_INSTS_MFMA SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT -d $OUT -o pmc_mfma --output-format csv -- $CMD > $OUT/pmc1.log 2>&1
rocprofv3 --pmc FETCH_SIZE -d $OUT -o pmc_fetch --output-format csv -- $CMD > $OUT/pmc2.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT -o pmc_write --output-format csv -- $CMD > $OUT/pmc3.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS -d $OUT -o pmc_grbm --output-format csv -- $CMD > $OUT/pmc4.log 2>&1
ls $OUT | head -20
