#!/bin/bash
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
( timeout 600 python tools/feeder_bench.py ) > $OUT/feeder.log 2>&1; echo "feeder rc=$?"
( timeout 900 python bench.py ) > $OUT/bench.log 2>&1; echo "bench rc=$?"
tail -3 $OUT/pytest_gpu.log; cat $OUT/feeder.log | tail -6; grep '^{' $OUT/bench.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['cpu_baseline'], d['parity'])"
