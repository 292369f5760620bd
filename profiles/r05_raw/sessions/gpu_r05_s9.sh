#!/bin/bash
# Round 5, session 9 (historical: the kernels it names were removed after it, profiles/EXPERIMENTS.md): the queued-walker sparse kernels (sparse_qw<L>_*): parity of every sparse variant on the small forests of the suite,
# then BASELINE config 4 -- the lock-step kernels against windows of 2 / 3 / 4 PU groups, both lane-state representations.
set -u
tag=${1:-r05_s9}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 600 python -m pytest tests/test_sparse.py -m gpu -x -q -k "every_sparse_kernel_variant" ) > $OUT/pytest_variants.log 2>&1; tail -5 $OUT/pytest_variants.log
( timeout 900 python tools/sparse_sweep.py --rows 4000000 --reps 3 --only sparse_dm1_k8_u8_t256,sparse_dk_k8_u8_t256,sparse_qw,sparse_qx --out $OUT/sparse_sweep.json ) > $OUT/sparse_sweep.log 2>&1; tail -14 $OUT/sparse_sweep.log
