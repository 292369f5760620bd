#!/bin/bash
# HEAD check at the end of the round's third session: full GPU suite, then the bench lines whose timing changed (no host synchronisation per step)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s43
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 175 python -m pytest tests -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/gpu_tests.log; grep -n "passed\|failed\|error" $OUT/gpu_tests.log | tail -3
( timeout 60 python bench.py ) > $OUT/bench_cfg3.log 2> $OUT/bench_cfg3.err; tail -1 $OUT/bench_cfg3.log | cut -c1-260
( timeout 25 python bench.py --config 2 --no-cpu-baseline --no-streamed ) > $OUT/bench_cfg2.log 2>/dev/null; tail -1 $OUT/bench_cfg2.log | cut -c1-200
( timeout 25 python bench.py --config 1 --no-cpu-baseline --no-streamed ) > $OUT/bench_cfg1.log 2>/dev/null; tail -1 $OUT/bench_cfg1.log | cut -c1-200
