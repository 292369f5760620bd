#!/bin/bash
# sparse kernels with dense level K (8-byte records only in LDS): K = 8 at two blocks per CU, K = 9 in one block -- A/B on the config-4 model
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s32
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 500 python tools/sparse_sweep.py --rows 4000000 --reps 3 --out $OUT/sparse_sweep_dk.json \
   --only sparse_k8_u8_t512,sparse_dk_k8_u8_t256,sparse_dk_k9_u8_t512,sparse_dk_k8_u8_t512,sparse_k7_u8_t256,sparse_dk_k7_u8_t256 ) 2>&1 | grep -v "^W\|amdgpu.ids" | cut -c1-200 | tee $OUT/sweep.log
( timeout 300 python bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed ) 2>&1 | grep -v "^W\|amdgpu.ids" | tail -1 | cut -c1-400 | tee $OUT/bench_cfg4.log
( timeout 900 python -m pytest tests/test_sparse.py tests/test_fuzz_gpu.py -m gpu -x -q 2>&1 | tail -15 ) > $OUT/tests.log; cat $OUT/tests.log
