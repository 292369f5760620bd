#!/bin/bash
# dense level K with 16 walks in flight per lane (two top passes through one image buffer, then one deep phase): A/B on the config-4 model
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s34
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 500 python tools/sparse_sweep.py --rows 4000000 --reps 3 --out $OUT/sparse_sweep_dk2.json \
   --only sparse_dk_k8_u8_t256,sparse_dk2_k8_u8_t256,sparse_dk2_k7_u8_t256,sparse_dk_k9_u8_t512,sparse_dk2_k9_u8_t512 ) 2>&1 | grep -v "^W\|amdgpu.ids" | cut -c1-200 | tee $OUT/sweep.log
( timeout 900 python -m pytest tests/test_sparse.py -m gpu -x -q 2>&1 | tail -15 ) > $OUT/tests.log; cat $OUT/tests.log
