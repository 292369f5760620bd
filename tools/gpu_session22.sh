#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | grep -v "Extension modules" | tail -3
timeout 600 python tools/sweep.py --shapes 8x4x16x20000000,16x4x28x20000000,32x6x16x10000000 --only stream --reps 5 --out gpurun_out/sweep_stream2.json > gpurun_out/s22_sweep.log 2>&1
grep -v "^/opt" gpurun_out/s22_sweep.log | tail -12
