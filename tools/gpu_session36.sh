#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | grep -v "Extension modules" | tail -4
timeout 600 python tools/sweep.py --shapes 500x10x28x10000000,500x9x28x10000000 --reps 3 --out gpurun_out/sweep_deep.json > gpurun_out/s36_sweep.log 2>&1
grep -v "^/opt" gpurun_out/s36_sweep.log | awk '{print $1,$2,$5,$6,$9,$10}' | tail -8
