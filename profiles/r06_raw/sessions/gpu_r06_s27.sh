#!/bin/bash
# Round 6, session 27: repeatability of the three-round shapes (lag on), three runs each.
set -u
tag=${1:-r06_s27}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
shape() { for o in 1 1 1 0; do ( DDT_SPARSE_R_LAG=$o timeout 300 python tools/run_shape.py --sparse --rows ${ROWS:-4000000} --reps 3 --opt sparse_r32=1 "$@" ) 2>&1 | tail -1 | cut -c1-150 | sed "s/^/[lag=$o $*] /"; done; }
shape --trees 512 --levels 14 --features 64 2>&1 | tee -a $OUT/sweep.log
shape --trees 512 --levels 15 --features 64 --full-levels 9 2>&1 | tee -a $OUT/sweep.log
shape --trees 256 --levels 15 --features 48 2>&1 | tee -a $OUT/sweep.log
