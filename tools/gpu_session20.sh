#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_q16.py tests/test_gpu_parity.py tests/test_multiclass.py -x -q -m gpu 2>&1 | grep -v "Extension modules" > gpurun_out/s20_tests.log
tail -6 gpurun_out/s20_tests.log
timeout 600 python tools/sweep.py --shapes 125x8x32x100000000 --only d8_ --reps 3 --out gpurun_out/sweep_f1.json > gpurun_out/s20_sweep.log 2>&1
timeout 600 python tools/sweep.py --shapes 125x8x32x100000000,100x6x28x10000000 --only q16 --reps 3 --opt q16_fused_prepass=0 --out gpurun_out/sweep_f0.json >> gpurun_out/s20_sweep.log 2>&1
timeout 600 python tools/sweep.py --shapes 100x6x28x10000000 --only d6_ --reps 5 --out gpurun_out/sweep_f2.json >> gpurun_out/s20_sweep.log 2>&1
grep -v "^/opt" gpurun_out/s20_sweep.log | grep -v "t512\|t256\|_reg\|dma " | tail -14
