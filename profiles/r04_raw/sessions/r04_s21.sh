#!/bin/bash
# the full-size periodic-batch tests (every row of every BASELINE config held to the oracle), then the whole GPU suite at HEAD
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s21; rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_full_size_gpu.py -q -x -m gpu --durations=10 2>&1 | grep -v "Extension modules" | tail -25 ) > $OUT/full_size.log; tail -16 $OUT/full_size.log
( timeout 1500 python -m pytest tests -q -m gpu --deselect tests/test_full_size_gpu.py 2>&1 | grep -v "Extension modules" | tail -8 ) > $OUT/gpu_tests.log; tail -3 $OUT/gpu_tests.log
