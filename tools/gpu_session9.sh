#!/bin/bash
# evidence for profiles/: rocprofv3 kernel-trace stats of bench.py + PMC passes on the rank-quantised path
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
rm -rf $OUT/prof_stats $OUT/pmcq1 $OUT/pmcq2 $OUT/pmcq3 $OUT/pmcq4
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"
SW="python $GRAFT_REPO_ROOT/tools/sweep.py --shapes 1000x8x32x8000000 --only q16_d8 --reps 2 --out /tmp/sw.json"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmcq1 -o pmc -- $SW ) > $OUT/pmcq1.log 2>&1; echo "pmc1 rc=$?"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM -d $OUT/pmcq2 -o pmc -- $SW ) > $OUT/pmcq2.log 2>&1; echo "pmc2 rc=$?"
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmcq3 -o pmc -- $SW ) > $OUT/pmcq3.log 2>&1; echo "pmc3 rc=$?"
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/pmcq4 -o pmc -- $SW ) > $OUT/pmcq4.log 2>&1; echo "pmc4 rc=$?"
