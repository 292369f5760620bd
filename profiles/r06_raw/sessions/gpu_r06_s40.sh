#!/bin/bash
# s40: the whole GPU suite + smoke at the commit with every cut form
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_s40
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/gpu_tests.log; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -3
( timeout 300 python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1; grep "smoke ok" $OUT/smoke.log | cut -c1-200
