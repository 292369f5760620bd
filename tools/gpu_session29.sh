#!/bin/bash
# PMC of the pre-pass kernels as they are at the end of the round (fused_rank_kernel on a 125-tree shard, rank_kernel on 1000 trees)
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/pmcF1 $OUT/pmcF2 $OUT/pmcR1 $OUT/pmcR2
SWF="python $GRAFT_REPO_ROOT/tools/sweep.py --shapes 125x8x32x16000000 --only q16_d8 --reps 2 --out /tmp/swf.json"
SWR="python $GRAFT_REPO_ROOT/tools/sweep.py --shapes 1000x8x32x16000000 --only q16_d8 --reps 2 --out /tmp/swr.json"
C1="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
C2="SQ_INSTS_VMEM SQ_ACTIVE_INST_VALU SQ_WAIT_INST_LDS SQ_WAIT_ANY GRBM_GUI_ACTIVE"
( cd /tmp && timeout 300 rocprofv3 --pmc $C1 -d $OUT/pmcF1 -o pmc -- $SWF ) > $OUT/pmcF1.log 2>&1; echo "F1 rc=$?"
( cd /tmp && timeout 300 rocprofv3 --pmc $C2 -d $OUT/pmcF2 -o pmc -- $SWF ) > $OUT/pmcF2.log 2>&1; echo "F2 rc=$?"
( cd /tmp && timeout 300 rocprofv3 --pmc $C1 -d $OUT/pmcR1 -o pmc -- $SWR ) > $OUT/pmcR1.log 2>&1; echo "R1 rc=$?"
( cd /tmp && timeout 300 rocprofv3 --pmc $C2 -d $OUT/pmcR2 -o pmc -- $SWR ) > $OUT/pmcR2.log 2>&1; echo "R2 rc=$?"
