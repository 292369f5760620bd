#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s14
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python tools/sweep.py --shapes 100x6x28x10000000,400x6x28x10000000,300x7x32x10000000 --only q16_d6,q16_d7 --reps 5 --out $OUT/sweep.json ) > $OUT/sweep.log 2>&1; grep -v "^W\|^E\|amdgpu.ids" $OUT/sweep.log | tail -14
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "6-28 or 6-64 or 7-20 or 7-32" 2>&1 | tail -4 ) > $OUT/tests.log; cat $OUT/tests.log
