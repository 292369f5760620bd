#!/bin/bash
# Round 6, session 14: which generic-bound perfect models of at most 64 tuple words gain on the sparse path (deep + fp64 sum, depth 16, PU groups beyond u16 ranks).
set -u
tag=${1:-r06_s14}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
shape() { for o in 1 0; do ( timeout 600 python tools/run_shape.py --rows ${ROWS:-4000000} --reps 2 --opt generic_via_sparse=$o "$@" ) 2>&1 | tail -1 | cut -c1-200 | sed "s/^/[via_sparse=$o $*] /"; done; }
shape --trees 512 --levels 12 --features 32 --sum-mode 1 2>&1 | tee -a $OUT/ab.log
shape --trees 512 --levels 12 --features 64 --sum-mode 1 2>&1 | tee -a $OUT/ab.log
shape --trees 256 --levels 10 --features 32 --sum-mode 1 2>&1 | tee -a $OUT/ab.log
shape --trees 256 --levels 9 --features 16 --sum-mode 1 2>&1 | tee -a $OUT/ab.log
shape --trees 64 --levels 15 --features 4 2>&1 | tee -a $OUT/ab.log
shape --trees 128 --levels 14 --features 64 --sum-mode 1 2>&1 | tee -a $OUT/ab.log
shape --trees 512 --levels 16 --features 64 --opt2 x 2>/dev/null | head -0
