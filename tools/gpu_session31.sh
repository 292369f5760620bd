#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_multiclass.py -x -q -m gpu -k "sharded_classifier" 2>&1 | grep -v "Extension modules" | tail -12
