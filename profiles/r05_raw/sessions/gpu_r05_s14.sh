#!/bin/bash
# Round 5, session 14: sparse kernels with dense PAIR records (sparse_dp_*): parity (the new GPU test + every sparse variant on the suite's forests),
# then BASELINE config 4 against the dense-mid kernel, alternating.
set -u
tag=${1:-r05_s14}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_sparse_dp.py tests/test_sparse.py -m gpu -x -q ) > $OUT/pytest_sparse.log 2>&1; tail -3 $OUT/pytest_sparse.log
( timeout 600 python tools/sparse_sweep.py --rows 4000000 --reps 3 --only sparse_dm1_k8_u8_t256,sparse_dp_k8_u8_t256,sparse_dp_k7_u8_t256,sparse_dk_k8_u8_t256 --out $OUT/sparse_sweep.json ) > $OUT/sparse_sweep.log 2>&1; grep variant $OUT/sparse_sweep.log | cut -c1-60,170-330
for rep in 1 2; do
  for dp in 1 0; do
    ( timeout 300 python bench.py --config 4 --no-streamed --no-cpu-baseline --opt sparse_dp=$dp ) > $OUT/bench_cfg4_dp${dp}_$rep.log 2>/dev/null; tail -1 $OUT/bench_cfg4_dp${dp}_$rep.log | cut -c100-200
  done
done
