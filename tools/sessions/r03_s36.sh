#!/bin/bash
# cluster-major image + one accumulator (q16_d8_c8_u4_gl_s2_cm) against the ring of 8 accumulators: A/B at 1000 trees, then the q16 / parity / adder GPU tests
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s36
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 400 python tools/sweep.py --shapes 1000x8x32x20000000 --only q16_d8_c8_u4_gl_s2 --out $OUT/sweep_q16_cm.json ) 2>&1 | grep -v "^W\|amdgpu.ids" | tail -6 | cut -c1-250 | tee $OUT/sweep.log
( timeout 900 python -m pytest tests/test_q16.py tests/test_adder_corner.py tests/test_gpu_parity.py tests/test_multiclass.py -m gpu -x -q 2>&1 | tail -12 ) > $OUT/tests.log; cat $OUT/tests.log
( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-streamed ) 2>&1 | grep -v "^W\|amdgpu.ids" | tail -1 | cut -c1-700 | tee $OUT/bench_cfg3.log
