#!/bin/bash
# Round 6, session 24: lagged sparse_r kernels for 2, 5, 6 rounds: A/B (DDT_SPARSE_R_LAG=1 / 0).
set -u
tag=${1:-r06_s24}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_sparse_r.py tests/test_sparse.py tests/test_zz_late_gpu.py -m gpu -x -q ) > $OUT/tests.log 2>&1; grep "passed\|failed" $OUT/tests.log
shape() { for o in 1 0; do ( DDT_SPARSE_R_LAG=$o timeout 300 python tools/run_shape.py --sparse --rows ${ROWS:-4000000} --reps 3 --opt sparse_r32=1 "$@" ) 2>&1 | tail -1 | cut -c1-220 | sed "s/^/[lag=$o $*] /"; done; }
shape --trees 256 --levels 12 --features 64 2>&1 | tee -a $OUT/sweep.log
shape --trees 512 --levels 12 --features 64 2>&1 | tee -a $OUT/sweep.log
shape --trees 512 --levels 13 --features 64 --full-levels 9 2>&1 | tee -a $OUT/sweep.log
shape --trees 300 --levels 18 --features 64 --full-levels 9 --permille 600 2>&1 | tee -a $OUT/sweep.log
shape --trees 300 --levels 20 --features 64 --full-levels 9 --permille 600 2>&1 | tee -a $OUT/sweep.log
shape --trees 512 --levels 10 --features 64 --full-levels 8 2>&1 | tee -a $OUT/sweep.log
