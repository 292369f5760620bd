#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/s17_tests.log 2>&1
tail -4 gpurun_out/s17_tests.log
timeout 600 python tools/classify_bench.py > gpurun_out/s17_classify.log 2>&1
grep -v "^/opt" gpurun_out/s17_classify.log | tail -4
timeout 300 python tools/sweep.py --shapes 100x6x28x10000000 --only d6_t1024 --reps 5 --out gpurun_out/sweep_u6.json > gpurun_out/s17_sweep.log 2>&1
grep -v "^/opt" gpurun_out/s17_sweep.log | tail -5
