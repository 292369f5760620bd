#!/bin/bash
# q16 with levels 0-1 from SGPRs: parity + A/B
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu > gpurun_out/s16_tests.log 2>&1
tail -5 gpurun_out/s16_tests.log
timeout 600 python tools/sweep.py --shapes 1000x8x32x100000000,500x8x32x100000000 --only q16_d8 --reps 3 --out gpurun_out/sweep_t.json > gpurun_out/s16_sweep.log 2>&1
grep -v "^/opt" gpurun_out/s16_sweep.log | tail -12
