#!/bin/bash
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 600 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
( timeout 600 python tools/sweep.py --shapes ${SHAPES:-1000x8x32x8000000} --reps 5 ) > $OUT/sweep.log 2>&1; echo "sweep rc=$?"
tail -5 $OUT/pytest_gpu.log; grep -v generic $OUT/sweep.log | tail -30
