#!/bin/bash
# profile session: rocprofv3 kernel-trace stats of bench.py + PMC passes on the dominant kernel
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 600 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
rocprofv3 -L > $OUT/counters_list.txt 2>&1
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof_stats -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > $OUT/prof_stats.log 2>&1; echo "stats rc=$?"
SW="python $GRAFT_REPO_ROOT/tools/sweep.py --shapes 1000x8x32x8000000 --only d8_t1024_r1_c4_u4_dma --reps 2 --out /tmp/sw.json"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $OUT/pmc1 -o pmc -- $SW ) > $OUT/pmc1.log 2>&1; echo "pmc1 rc=$?"
( cd /tmp && timeout 300 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL -d $OUT/pmc2 -o pmc -- $SW ) > $OUT/pmc2.log 2>&1; echo "pmc2 rc=$?"
( cd /tmp && timeout 300 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc3 -o pmc -- $SW ) > $OUT/pmc3.log 2>&1; echo "pmc3 rc=$?"
( cd /tmp && timeout 300 rocprofv3 --pmc WRITE_SIZE GRBM_GUI_ACTIVE -d $OUT/pmc4 -o pmc -- $SW ) > $OUT/pmc4.log 2>&1; echo "pmc4 rc=$?"
find $OUT -name "*.csv" | head -30
tail -3 $OUT/pytest_gpu.log
