#!/bin/bash
# Round 5, session 19: dense pair records in the rank-quantised sparse family (sparse_qp_*): parity (the pair-record tests + the whole sparse
# suite: the automatic choice changed for forests that fit u16 ranks), then a 255-bin version of the config-4 forest, qp against qd.
set -u
tag=${1:-r05_s19}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1200 python -m pytest tests/test_sparse_dp.py tests/test_sparse.py tests/test_sparse_dm.py tests/test_importer.py -m gpu -x -q ) > $OUT/pytest_sparse.log 2>&1; tail -3 $OUT/pytest_sparse.log
for F in 64 32; do
  ( timeout 300 python tools/run_shape.py --sparse --trees 512 --levels 16 --features $F --rows 10000000 --bins 255 --reps 3 --opt sparse_dp=-1 ) 2>&1 | tail -2 | cut -c1-250
  ( timeout 300 python tools/run_shape.py --sparse --trees 512 --levels 16 --features $F --rows 10000000 --bins 255 --reps 3 --opt sparse_dp=0 ) 2>&1 | tail -2 | cut -c1-250
done
