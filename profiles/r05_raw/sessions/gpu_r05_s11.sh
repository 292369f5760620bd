#!/bin/bash
# Round 5, session 11: sparse kernels -- the last possible round of the deep loop without its gather (option sparse_peel_last): parity of the sparse
# suite, then BASELINE config 4 A/B, alternating, three pairs.
set -u
tag=${1:-r05_s11}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_sparse.py tests/test_sparse_dm.py -m gpu -x -q ) > $OUT/pytest_sparse.log 2>&1; tail -3 $OUT/pytest_sparse.log
for rep in 1 2 3; do
  ( timeout 300 python bench.py --config 4 --no-streamed --no-cpu-baseline --opt sparse_peel_last=1 ) > $OUT/bench_cfg4_peel1_$rep.log 2>/dev/null; tail -1 $OUT/bench_cfg4_peel1_$rep.log | cut -c1-200
  ( timeout 300 python bench.py --config 4 --no-streamed --no-cpu-baseline --opt sparse_peel_last=0 ) > $OUT/bench_cfg4_peel0_$rep.log 2>/dev/null; tail -1 $OUT/bench_cfg4_peel0_$rep.log | cut -c1-200
done
