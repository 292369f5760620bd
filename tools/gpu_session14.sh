#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -x -q -m gpu > gpurun_out/s14_tests.log 2>&1
tail -4 gpurun_out/s14_tests.log
timeout 600 python bench.py --steps 5 --warmup 2 > gpurun_out/s14_bench.log 2>&1
tail -2 gpurun_out/s14_bench.log
