#!/bin/bash
# rank-quantised sparse kernels on a forest whose thresholds fit 16-bit ranks (255 / 4096 values per feature) + sparse GPU parity
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s27
rm -rf "$OUT"; mkdir -p "$OUT"
for b in 255 4096; do for o in "sparse_q16=1" "sparse_q16=0"; do
  echo "== bins $b $o"
  ( timeout 300 python tools/run_shape.py --sparse --trees 512 --levels 16 --features 64 --rows 10000000 --reps 3 --bins $b --opt $o ) 2>&1 | grep -v "^W\|amdgpu.ids" | tail -2
done; done | tee $OUT/cfg4_bins.log
( timeout 1500 python -m pytest tests -m gpu -q -k "sparse" 2>&1 | tail -8 ) > $OUT/tests.log; cat $OUT/tests.log
