#!/bin/bash
# Round 6, session 5: where the sparse_r family wins: shapes x {sparse_r32 = 1, 0}; then the whole sparse GPU suite with the automatic rule on.
set -u
tag=${1:-r06_s20}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
shape() { for o in 1 0; do ( timeout 300 python tools/run_shape.py --sparse --rows ${ROWS:-4000000} --reps 3 --opt sparse_r32=$o "$@" ) 2>&1 | tail -1 | cut -c1-220 | sed "s/^/[r32=$o $*] /"; done; }
shape --trees 512 --levels 16 --features 64 2>&1 | tee -a $OUT/sweep.log
shape --trees 512 --levels 16 --features 64 --bins 255 2>&1 | tee -a $OUT/sweep.log
shape --trees 512 --levels 16 --features 32 2>&1 | tee -a $OUT/sweep.log
shape --trees 512 --levels 16 --features 128 2>&1 | tee -a $OUT/sweep.log
shape --trees 256 --levels 12 --features 64 2>&1 | tee -a $OUT/sweep.log
shape --trees 128 --levels 14 --features 20 --full-levels 6 2>&1 | tee -a $OUT/sweep.log
shape --trees 64 --levels 16 --features 64 2>&1 | tee -a $OUT/sweep.log
shape --trees 512 --levels 20 --features 64 --full-levels 8 --permille 800 2>&1 | tee -a $OUT/sweep.log
shape --trees 512 --levels 10 --features 64 --full-levels 8 2>&1 | tee -a $OUT/sweep.log
shape --trees 1000 --levels 13 --features 28 --full-levels 7 --permille 750 2>&1 | tee -a $OUT/sweep.log
