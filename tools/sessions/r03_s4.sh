#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s4
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 600 python -m pytest tests/test_q16.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -5 ) > $OUT/tests.log; cat $OUT/tests.log
( timeout 600 python tools/sweep.py --shapes 1000x8x32x100000000 --only q16_d8 --reps 3 --out $OUT/sweep_q16_d8.json ) > $OUT/sweep.log 2>&1; grep -v "^W\|^E" $OUT/sweep.log | tail -5
