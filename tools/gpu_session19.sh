#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "odd_shapes" 2>&1 | grep -v "Extension modules" | tail -3
timeout 900 python tools/sweep.py --shapes 512x16x64x1000000,64x12x64x1000000 --reps 2 --out gpurun_out/sweep_cfg4.json > gpurun_out/s19_sweep.log 2>&1
grep -v "^/opt" gpurun_out/s19_sweep.log | tail -5
