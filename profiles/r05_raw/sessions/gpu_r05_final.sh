#!/bin/bash
# Round 5: the GPU suite + smoke + the driver's command at HEAD (last call of the round).
set -u
tag=${1:-r05_final}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/gpu_tests.log; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -2
( timeout 300 python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1; grep "smoke ok" $OUT/smoke.log
( timeout 900 python bench.py ) > $OUT/bench_default.log 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.log | cut -c1-300
