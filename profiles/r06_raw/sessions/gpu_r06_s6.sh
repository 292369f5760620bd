#!/bin/bash
# Round 6, session 6: the sparse_r family's GPU tests + the sparse suites under the refined automatic rule; config 4 with its profile.
set -u
tag=${1:-r06_s6}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1200 python -m pytest tests/test_sparse_r.py tests/test_sparse.py tests/test_sparse_dp.py tests/test_sparse_dm.py tests/test_importer.py -m gpu -q --durations=5 ) > $OUT/pytest_sparse.log 2>&1; tail -15 $OUT/pytest_sparse.log
( timeout 300 python bench.py --config 4 --no-cpu-baseline --no-other-configs --no-other-modes --no-streamed ) > $OUT/bench_cfg4.log 2>&1; tail -1 $OUT/bench_cfg4.log | cut -c1-400
