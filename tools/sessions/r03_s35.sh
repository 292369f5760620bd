#!/bin/bash
# HEAD check (GPU suite, smoke, the default bench line) + the coalesced-load microbenchmark
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s35
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 200 tools/ubench/ubench coal ) > $OUT/ubench_coal.json 2> $OUT/ubench_coal.err; echo "ubench rc=$?"; cat $OUT/ubench_coal.json
( timeout 1200 python -m pytest tests -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/gpu_tests.log; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -2
( timeout 300 python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1; grep "smoke ok" $OUT/smoke.log
( timeout 900 python bench.py ) > $OUT/bench_cfg3.log 2> $OUT/bench_cfg3.err; tail -1 $OUT/bench_cfg3.log | cut -c1-300
