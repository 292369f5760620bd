#!/bin/bash
# stream kernel with skewed feature rows (conflict-free transposed stores) + parity
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s20
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python tools/sweep.py --shapes 8x4x16x200000000,8x4x16x50000000 --only stream_d4_u4_l4 --reps 7 --out $OUT/sweep.json ) > $OUT/sweep.log 2>&1; grep -v "^W\|^E\|amdgpu.ids" $OUT/sweep.log | tail -16
( timeout 300 python tools/run_shape.py --trees 8 --levels 4 --features 32 --rows 100000000 --reps 5 ) 2>&1 | grep -v "^W\|^E\|amdgpu.ids" | tail -1 | tee -a $OUT/cfg1.log
( timeout 300 python tools/run_shape.py --trees 8 --levels 4 --features 12 --rows 100000000 --reps 5 ) 2>&1 | grep -v "^W\|^E\|amdgpu.ids" | tail -1 | tee -a $OUT/cfg1.log
( timeout 300 python tools/run_shape.py --trees 30 --levels 6 --features 16 --rows 100000000 --reps 5 --variant stream_d6_u4_l4 ) 2>&1 | grep -v "^W\|^E\|amdgpu.ids" | tail -1 | tee -a $OUT/cfg1.log
( timeout 900 python -m pytest tests -m gpu -x -q -k "stream or parity or variant or crafted or fuzz" 2>&1 | tail -5 ) > $OUT/tests.log; cat $OUT/tests.log
