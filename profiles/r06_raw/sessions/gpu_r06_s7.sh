#!/bin/bash
# Round 6, session 7: the whole GPU suite after the pruning, the deep kernels' peeled first wait and the sparse_r family; configs 3 / 6 / shard unchanged?
set -u
tag=${1:-r06_s7}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1500 python -m pytest tests -m gpu -x -q --durations=8 ) > $OUT/gpu_tests.log 2>&1; tail -14 $OUT/gpu_tests.log
for c in "--config 6" "--config 3" "--shard-of 8"; do
  n=$(echo $c | tr -d ' -')
  ( timeout 300 python bench.py $c --no-cpu-baseline --no-other-configs --no-other-modes --no-streamed ) > $OUT/bench_$n.log 2>&1
  tail -1 $OUT/bench_$n.log | cut -c1-260
done
