#!/bin/bash
# rocprofv3 counter-only passes over the secondary kernels (tools/secondary_prof.py)
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/prof_secondary_pmc
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
CMD="python $R/tools/secondary_prof.py"
rocprofv3 --pmc FETCH_SIZE -d $OUT -o pmc_fetch --output-format csv -- $CMD > $OUT/p1.log 2>&1
rocprofv3 --pmc WRITE_SIZE -d $OUT -o pmc_write --output-format csv -- $CMD > $OUT/p2.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_BUSY_CU_CYCLES SQ_INSTS_VALU SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT -o pmc_sq --output-format csv -- $CMD > $OUT/p3.log 2>&1
ls $OUT
