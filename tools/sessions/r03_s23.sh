#!/bin/bash
# q16 kernels: two alternating SGPR sets for the _s2 top records (no copies) + the hot path as straight-line code (_h)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s23
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 600 python tools/sweep.py --shapes 100x6x28x10000000 --only q16_d6_c16_u4_s2 --reps 7 --out $OUT/sweep_d6.json ) > $OUT/sweep_d6.log 2>&1; echo "rc=$?"; grep -v "^W\|amdgpu.ids" $OUT/sweep_d6.log | tail -4
( timeout 600 python tools/sweep.py --shapes 1000x8x32x20000000 --only q16_d8_c8_u4_gl_s2 --reps 5 --out $OUT/sweep_d8.json ) > $OUT/sweep_d8.log 2>&1; echo "rc=$?"; grep -v "^W\|amdgpu.ids" $OUT/sweep_d8.log | tail -4
( timeout 900 python -m pytest tests/test_q16.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -4 ) > $OUT/tests.log; cat $OUT/tests.log
