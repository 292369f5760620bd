#!/bin/bash
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
SW="python $GRAFT_REPO_ROOT/tools/sweep.py --shapes 1000x8x32x8000000 --only q16_d8 --reps 2 --out /tmp/sw.json"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY -d $OUT/pmcq1 -o pmc -- $SW ) > $OUT/pmcq1.log 2>&1; echo "pmc1 rc=$?"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE SQ_WAVES -d $OUT/pmcq2 -o pmc -- $SW ) > $OUT/pmcq2.log 2>&1; echo "pmc2 rc=$?"
