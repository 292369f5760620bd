#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "config4_shape" 2>&1 | grep -v "Extension modules" | tail -15
