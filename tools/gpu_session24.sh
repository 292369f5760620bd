#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_q16.py tests/test_gpu_parity.py tests/test_multiclass.py -x -q -m gpu 2>&1 | grep -v "Extension modules" | tail -3
timeout 900 python tools/sweep.py --shapes 1000x8x32x100000000,125x8x32x100000000,100x6x28x10000000,64x4x32x10000000 --only q16 --reps 3 --out gpurun_out/sweep_pair.json > gpurun_out/s24_sweep.log 2>&1
grep -v "^/opt" gpurun_out/s24_sweep.log | tail -6
