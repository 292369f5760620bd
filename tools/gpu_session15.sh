#!/bin/bash
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_q16.py tests/test_gpu_parity.py -x -q -m gpu > gpurun_out/s15_tests.log 2>&1
tail -4 gpurun_out/s15_tests.log
timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/s15_bench.log 2>&1
tail -1 gpurun_out/s15_bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['kernel_ms'], d['roofline']['prepass_ms'])"
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/s15_prof -o s15 -- python $GRAFT_REPO_ROOT/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/s15_prof.log 2>&1
cd $GRAFT_REPO_ROOT; find gpurun_out/s15_prof -name "*kernel_stats.csv" | head -1 | xargs head -6
