#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "Extension modules" > gpurun_out/s21_tests.log
tail -4 gpurun_out/s21_tests.log
for o in 1 0; do
timeout 900 python tools/sweep.py --shapes 125x8x32x100000000,250x8x32x100000000,500x8x32x100000000 --only q16 --reps 3 --opt q16_fused_prepass=$o --out gpurun_out/sweep_g$o.json > gpurun_out/s21_sweep$o.log 2>&1
echo "fused=$o"; grep -v "^/opt" gpurun_out/s21_sweep$o.log | tail -3
done
