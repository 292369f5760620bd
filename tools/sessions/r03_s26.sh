#!/bin/bash
# rank-quantised sparse kernels (sparse_q_*): config 4 against the fp32-tile kernels + GPU parity of the sparse path
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s26
rm -rf "$OUT"; mkdir -p "$OUT"
for o in "sparse_q16=1" "sparse_q16=0" "sparse_q16=1,sparse_top_levels=7" "sparse_q16=1,sparse_top_levels=6"; do
  echo "== $o"
  ( timeout 300 python tools/run_shape.py --sparse --trees 512 --levels 16 --features 64 --rows 10000000 --reps 3 --opt $o ) 2>&1 | grep -v "^W\|amdgpu.ids" | tail -2
done | tee $OUT/cfg4.log
( timeout 300 python tools/run_shape.py --sparse --trees 512 --levels 16 --features 32 --rows 10000000 --reps 3 --opt sparse_q16=1 ) 2>&1 | grep -v "^W\|amdgpu.ids" | tail -1 | tee -a $OUT/cfg4.log
( timeout 300 python tools/run_shape.py --sparse --trees 512 --levels 16 --features 32 --rows 10000000 --reps 3 --opt sparse_q16=0 ) 2>&1 | grep -v "^W\|amdgpu.ids" | tail -1 | tee -a $OUT/cfg4.log
( timeout 1200 python -m pytest tests -m gpu -x -q -k "sparse" 2>&1 | tail -6 ) > $OUT/tests.log; cat $OUT/tests.log
