#!/bin/bash
# Round 6, session 2: the "sparse_r_*" family (32-bit ranks, pair records on every deep level): parity of every variant, then BASELINE config 4 A/B.
set -u
tag=${1:-r06_s2}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_sparse.py -m gpu -x -q -k "every_sparse_kernel_variant" ) > $OUT/pytest_variants.log 2>&1; tail -5 $OUT/pytest_variants.log
for o in "sparse_r32=-1" "sparse_r32=0"; do
  ( timeout 300 python bench.py --config 4 --opt $o --no-cpu-baseline --no-other-configs --no-other-modes --no-streamed ) > $OUT/bench_cfg4_$o.log 2>&1
  tail -1 $OUT/bench_cfg4_$o.log | cut -c1-900
done
( timeout 900 python -m pytest tests/test_sparse.py tests/test_sparse_dp.py tests/test_sparse_dm.py tests/test_importer.py -m gpu -x -q ) > $OUT/pytest_sparse.log 2>&1; tail -5 $OUT/pytest_sparse.log
