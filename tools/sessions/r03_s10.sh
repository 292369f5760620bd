#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s10
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python tools/sweep.py --shapes 8x4x16x200000000,8x4x32x100000000,16x4x16x100000000 --only stream_d4 --reps 5 --out $OUT/sweep_stream.json ) > $OUT/sweep.log 2>&1; grep -v "^W\|^E\|amdgpu.ids" $OUT/sweep.log | tail -30
( timeout 900 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "8-4-16 or 16-4-8 or 40-4-200 or 200-3-12" 2>&1 | tail -5 ) > $OUT/tests.log; cat $OUT/tests.log
