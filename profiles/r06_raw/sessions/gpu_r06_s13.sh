#!/bin/bash
# Round 6, session 13: perfect-tree models without a tuned kernel through the sparse-forest path: the whole GPU suite, then what it buys on the shapes that fell to `generic`.
set -u
tag=${1:-r06_s13}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1800 python -m pytest tests -m gpu -x -q --durations=6 ) > $OUT/gpu_tests.log 2>&1; tail -12 $OUT/gpu_tests.log
shape() { for o in 1 0; do ( timeout 600 python tools/run_shape.py --rows ${ROWS:-4000000} --reps 2 --opt generic_via_sparse=$o "$@" ) 2>&1 | tail -1 | cut -c1-200 | sed "s/^/[via_sparse=$o $*] /"; done; }
shape --trees 512 --levels 12 --features 200 2>&1 | tee -a $OUT/ab.log
shape --trees 512 --levels 12 --features 100 2>&1 | tee -a $OUT/ab.log
shape --trees 256 --levels 9 --features 400 2>&1 | tee -a $OUT/ab.log
shape --trees 512 --levels 16 --features 64 2>&1 | tee -a $OUT/ab.log
shape --trees 512 --levels 16 --features 32 2>&1 | tee -a $OUT/ab.log
shape --trees 64 --levels 15 --features 200 2>&1 | tee -a $OUT/ab.log
