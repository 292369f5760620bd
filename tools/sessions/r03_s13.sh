#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s13
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_sparse.py tests/test_adder_corner.py -m gpu -x -q 2>&1 | tail -8 ) > $OUT/tests.log; cat $OUT/tests.log
( timeout 900 python tools/sparse_sweep.py --rows 4000000 --only k8_u8_t512,k7_u8_t512,k8_u8_t256,k9_u8_t256 --out $OUT/sparse_sweep.json ) > $OUT/sweep.log 2>&1; grep -v "^W\|^E\|amdgpu.ids" $OUT/sweep.log | tail -14
