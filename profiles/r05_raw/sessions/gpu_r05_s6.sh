#!/bin/bash
# Round 5, session 6: config 4 -- K = 7 / 8 with and without a dense mid level, both deep-record orders (forced variants, out-of-range idle gathers).
set -u
tag=${1:-r05_s6}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python tools/sparse_sweep.py --rows 4000000 --orders 0,1 --only sparse_dm1_k8_u8_t256,sparse_dm1_k7_u8_t256,sparse_dk_k8_u8_t256,sparse_dk_k7_u8_t256,sparse_dm2_k8_u8_t256 --out $OUT/sparse_sweep.json ) > $OUT/sparse_sweep.log 2>&1; tail -14 $OUT/sparse_sweep.log
