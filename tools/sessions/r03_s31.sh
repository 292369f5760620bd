#!/bin/bash
# sparse kernel: pair prefetch in the top phase (one LDS round trip per level), top-phase-only timing ablation; sparse GPU tests on the shipped r1 form
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s31
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 500 python tools/sparse_sweep.py --rows 4000000 --reps 3 --out $OUT/sparse_sweep_pf.json \
   --only sparse_k8_u8_t512,sparse_pf_k8_u8_t512,sparse_topo_k8_u8_t512,sparse_k7_u8_t256,sparse_pf_k7_u8_t256,sparse_topo_k7_u8_t256 ) 2>&1 | grep -v "^W\|amdgpu.ids" | cut -c1-200 | tee $OUT/sweep.log
for v in sparse_q_k8_u8_t1024 sparse_qpf_k8_u8_t1024; do
  echo "== bins 255 $v"
  ( timeout 200 python tools/run_shape.py --sparse --trees 512 --levels 16 --features 64 --rows 10000000 --reps 3 --bins 255 --variant $v ) 2>&1 | grep -v "^W\|amdgpu.ids" | tail -1
done | tee $OUT/cfg4_bins.log
( timeout 700 python -m pytest tests/test_sparse.py -m gpu -x -q 2>&1 | tail -15 ) > $OUT/tests.log; cat $OUT/tests.log
