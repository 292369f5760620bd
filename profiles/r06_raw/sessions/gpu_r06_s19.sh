#!/bin/bash
# Round 6, session 19: sparse_r with TWO PU groups per pass (16 chains per lane, K = 8): parity of the variant, then config 4 forced against the default.
set -u
tag=${1:-r06_s19}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_sparse.py -m gpu -q -x -k "every_sparse_kernel_variant" ) > $OUT/pytest.log 2>&1; tail -3 $OUT/pytest.log
for v in sparse_r_k8_u16_t256 sparse_r_k9_u8_t256 sparse_r_k8_u8_t256 sparse_r_k8_u16_t256 sparse_r_k9_u8_t256; do
  ( timeout 300 python tools/run_shape.py --sparse --trees 512 --levels 16 --features 64 --rows 10000000 --reps 3 --variant $v ) 2>&1 | tail -1 | cut -c1-200
done
for v in sparse_r_k8_u16_t256 sparse_r_k10_u8_t256; do
  ( timeout 300 python tools/run_shape.py --sparse --trees 512 --levels 16 --features 32 --rows 10000000 --reps 3 --variant $v ) 2>&1 | tail -1 | cut -c1-200
done
