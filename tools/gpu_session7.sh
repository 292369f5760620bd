#!/bin/bash
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/prof_q16 -o q16 -- python $GRAFT_REPO_ROOT/tools/sweep.py --shapes ${SHAPES:-1000x8x32x8000000} --only ${ONLY:-q16} --reps 5 --out /tmp/sw.json ) > $OUT/prof_q16.log 2>&1; echo "prof rc=$?"
tail -3 $OUT/pytest_gpu.log; grep -E "q16|ok=" $OUT/prof_q16.log | tail -5
