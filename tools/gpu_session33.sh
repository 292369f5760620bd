#!/bin/bash
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_q16.py -x -q -m gpu 2>&1 | grep -v "Extension modules" | tail -2
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['prepass_ms'])"
