#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --force-collectives --trees 125 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('125 trees pipeline', d['ms_per_step'])"
timeout 600 python bench.py --steps 4 --warmup 1 --no-cpu-baseline --trees 125 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('125 trees plain', d['ms_per_step'])"
timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --force-collectives 2>&1 | grep '"metric"' | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('1000 trees pipeline', d['ms_per_step'])"
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_multiclass.py tests/test_rowshard.py -x -q -m gpu -k "rank or shard" 2>&1 | grep -v "Extension modules" | tail -2
