#!/bin/bash
mkdir -p gpurun_out
timeout 600 python tools/sweep.py --shapes 500x8x400x2000000,500x8x200x5000000 --reps 3 --out gpurun_out/sweep_wide.json > gpurun_out/s37_sweep.log 2>&1
grep -v "^/opt" gpurun_out/s37_sweep.log | awk '{print $1,$2,$5,$6,$9,$10}' | tail -20
