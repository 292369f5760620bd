#!/bin/bash
# two-level blocks (sparse_b2_*): GPU parity of the sparse suite, then the config-4 model through b2 vs one record per node
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s28
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 700 python -m pytest tests/test_sparse.py -m gpu -x -q 2>&1 | tail -15 ) > $OUT/tests.log; cat $OUT/tests.log
( timeout 400 python tools/sparse_sweep.py --rows 4000000 --reps 3 --out $OUT/sparse_sweep_b2.json \
   --only sparse_k8_u8_t512,sparse_b2_k8_u8_t512,sparse_b2_k9_u8_t512,sparse_b2_k7_u8_t512,sparse_b2_k8_u8_t256,sparse_b2_k9_u8_t256,sparse_b2_k10_u8_t256 ) 2>&1 | grep -v "^W\|amdgpu.ids" | tee $OUT/sweep.log
for o in "sparse_q16=1" "sparse_q16=0" "sparse_q16=0,sparse_b2=0"; do
  echo "== bins 255 $o"
  ( timeout 200 python tools/run_shape.py --sparse --trees 512 --levels 16 --features 64 --rows 10000000 --reps 3 --bins 255 --opt $o ) 2>&1 | grep -v "^W\|amdgpu.ids" | tail -2
done | tee $OUT/cfg4_bins.log
( timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed ) 2>&1 | grep -v "^W\|amdgpu.ids" | tail -2 | tee $OUT/bench_cfg4.log
