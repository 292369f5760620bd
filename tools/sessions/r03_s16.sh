#!/bin/bash
# stream kernel after the VALU trim (FIXED line count, scalar missing mask, total_ring) + occupancy-sized grid
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s16
rm -rf "$OUT"; mkdir -p "$OUT"
for o in "" "stream_blocks_per_cu=8" "stream_blocks_per_cu=5" "stream_blocks_per_cu=10"; do
  echo "== opt=$o"
  ( timeout 300 python tools/run_shape.py --trees 8 --levels 4 --features 16 --rows 200000000 --reps 5 ${o:+--opt $o} ) 2>&1 | grep -v "^W\|^E\|amdgpu.ids" | tail -2
done | tee $OUT/cfg1.log
( timeout 300 python tools/run_shape.py --trees 8 --levels 4 --features 32 --rows 100000000 --reps 5 ) 2>&1 | grep -v "^W\|^E\|amdgpu.ids" | tail -1 | tee -a $OUT/cfg1.log
( timeout 300 python tools/run_shape.py --trees 8 --levels 4 --features 12 --rows 100000000 --reps 5 ) 2>&1 | grep -v "^W\|^E\|amdgpu.ids" | tail -1 | tee -a $OUT/cfg1.log
( timeout 900 python -m pytest tests -m gpu -x -q -k "stream or parity or variant or crafted" 2>&1 | tail -5 ) > $OUT/tests.log; cat $OUT/tests.log
