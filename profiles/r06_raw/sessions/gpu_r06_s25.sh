#!/bin/bash
# Round 6, session 25: the whole GPU suite + config 4 / headline after the lagged sparse_r kernels.
set -u
tag=${1:-r06_s25}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1500 python -m pytest tests -m gpu -x -q ) > $OUT/gpu_tests.log 2>&1; grep "passed\|failed" $OUT/gpu_tests.log
for i in 1 2 3; do ( timeout 600 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-streamed ) 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('cfg4', d['value'], d['ms_per_step'], r['kernel'], r['kernel_ms'], r['prepass_ms'], d.get('parity'))" | tee -a $OUT/cfg4.log; done
( timeout 900 python bench.py ) > $OUT/bench_default.log 2>$OUT/bench_default.err; tail -1 $OUT/bench_default.log | cut -c1-300
python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; tail -1 $OUT/smoke.log
