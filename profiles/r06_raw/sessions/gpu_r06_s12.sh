#!/bin/bash
# Round 6, session 12: feature compaction -- wide perfect-tree models that test <= 64 features on the rank-quantised kernels: parity + A/B against the kernels they ran on.
set -u
tag=${1:-r06_s12}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_q16_deep.py -m gpu -q -x ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
shape() { for o in 1 0; do ( timeout 300 python tools/run_shape.py --rows ${ROWS:-10000000} --reps 3 --opt feature_compaction=$o "$@" ) 2>&1 | tail -1 | cut -c1-200 | sed "s/^/[compaction=$o $*] /"; done; }
shape --trees 512 --levels 12 --features 60 --wide-features 200 2>&1 | tee -a $OUT/ab.log
shape --trees 512 --levels 12 --features 32 --wide-features 200 2>&1 | tee -a $OUT/ab.log
shape --trees 256 --levels 9 --features 64 --wide-features 1000 2>&1 | tee -a $OUT/ab.log
shape --trees 1000 --levels 8 --features 48 --wide-features 132 2>&1 | tee -a $OUT/ab.log
shape --trees 1000 --levels 8 --features 32 --wide-features 100 2>&1 | tee -a $OUT/ab.log
