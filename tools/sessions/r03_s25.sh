#!/bin/bash
# q16 _gl kernels: leaf gathers through a buffer resource (no VALU address arithmetic)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s25
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 600 python tools/sweep.py --shapes 1000x8x32x20000000,125x8x32x20000000 --only q16_d8_c8_u4_gl --reps 5 --out $OUT/sweep_d8.json ) > $OUT/sweep_d8.log 2>&1; echo "rc=$?"; grep -v "^W\|amdgpu.ids" $OUT/sweep_d8.log | tail -4
( timeout 900 python -m pytest tests/test_q16.py tests/test_gpu_parity.py tests/test_adder_corner.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -4 ) > $OUT/tests.log; cat $OUT/tests.log
