#!/bin/bash
# deep-phase schedules (two-phase / rotating, 1 or 2 visits per step) for the classic and the two-level-block sparse kernels; lean block visit
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s30
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 500 python tools/sparse_sweep.py --rows 4000000 --reps 3 --out $OUT/sparse_sweep_rot.json \
   --only sparse_k8_u8_t512,sparse_r1_k8_u8_t512,sparse_r2_k8_u8_t512,sparse_b2_k8_u8_t512,sparse_b2r1_k8_u8_t512,sparse_b2r2_k8_u8_t512,sparse_k7_u8_t256,sparse_r1_k7_u8_t256,sparse_r2_k7_u8_t256 ) 2>&1 | grep -v "^W\|amdgpu.ids" | cut -c1-200 | tee $OUT/sweep.log
for v in sparse_q_k8_u8_t1024 sparse_qr1_k8_u8_t1024 sparse_qr2_k8_u8_t1024; do
  echo "== bins 255 $v"
  ( timeout 200 python tools/run_shape.py --sparse --trees 512 --levels 16 --features 64 --rows 10000000 --reps 3 --bins 255 --variant $v ) 2>&1 | grep -v "^W\|amdgpu.ids" | tail -1
done | tee $OUT/cfg4_bins.log
( timeout 700 python -m pytest tests/test_sparse.py -m gpu -x -q 2>&1 | tail -15 ) > $OUT/tests.log; cat $OUT/tests.log
