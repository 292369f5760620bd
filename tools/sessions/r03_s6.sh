#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s6
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1500 python -m pytest tests -m gpu -q --durations=10 2>&1 | grep -v "Extension modules" | tail -40 ) > $OUT/gpu_tests.log; tail -25 $OUT/gpu_tests.log
