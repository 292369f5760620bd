#!/bin/bash
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 300 python tools/sweep.py --shapes 1000x8x32x8000000 --only d8_t1024 --reps 5 ) > $OUT/sweep.log 2>&1; echo "sweep rc=$?"
for V in d8_t1024_r1_c4_u4_dma_la d8_t1024_r1_c4_u4_dma_t d8_t1024_r1_c4_u4_dma_abl8; do
SW="python $GRAFT_REPO_ROOT/tools/sweep.py --shapes 1000x8x32x8000000 --only $V --reps 2 --out /tmp/sw.json"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_WAIT_ANY -d $OUT/pmcA_$V -o pmc -- $SW ) > $OUT/pmcA_$V.log 2>&1; echo "pmcA $V rc=$?"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE -d $OUT/pmcB_$V -o pmc -- $SW ) > $OUT/pmcB_$V.log 2>&1; echo "pmcB $V rc=$?"
done
grep -v generic $OUT/sweep.log | tail -12
