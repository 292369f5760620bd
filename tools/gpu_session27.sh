#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_fuzz_gpu.py -x -q -m gpu 2>&1 | grep -v "Extension modules" | tail -15
