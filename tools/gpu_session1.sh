#!/bin/bash
# first GPU session: smoke, parity tests, variant sweep, bench, rocprof stats
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
( time timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > gpurun_out/smoke.log 2>&1
echo "smoke rc=$?" | tee -a gpurun_out/smoke.log
( time timeout 900 python -m pytest tests -m gpu -x -q ) > gpurun_out/pytest_gpu.log 2>&1
echo "pytest rc=$?" | tee -a gpurun_out/pytest_gpu.log
( time timeout 600 python tools/sweep.py ) > gpurun_out/sweep.log 2>&1
echo "sweep rc=$?" | tee -a gpurun_out/sweep.log
( time timeout 600 python bench.py --steps 3 --warmup 1 ) > gpurun_out/bench.log 2>&1
echo "bench rc=$?" | tee -a gpurun_out/bench.log
tail -3 gpurun_out/smoke.log; tail -5 gpurun_out/pytest_gpu.log; tail -40 gpurun_out/sweep.log; tail -3 gpurun_out/bench.log
