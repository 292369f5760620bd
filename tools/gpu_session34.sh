#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "Extension modules" | tail -12
timeout 600 python tools/sweep.py --shapes 500x7x28x10000000,500x5x28x10000000,500x3x16x10000000 --reps 3 --out gpurun_out/sweep_odd.json > gpurun_out/s34_sweep.log 2>&1
grep -v "^/opt" gpurun_out/s34_sweep.log | awk '{print $1,$2,$5,$6,$9,$10}' | tail -30
