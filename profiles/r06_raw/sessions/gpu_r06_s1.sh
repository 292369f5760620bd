#!/bin/bash
# Round 6, session 1: first contact with a real peer (two ranks on ONE device through RCCL), and this box's baselines of the numbers the round works on.
set -u
tag=${1:-r06_s1}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 600 python -m pytest tests/test_comm_two_ranks_gpu.py -m gpu -x -q -rs ) > $OUT/two_rank.log 2>&1; tail -15 $OUT/two_rank.log
cp gpurun_out/two_rank_probe.json $OUT/ 2>/dev/null
for c in "--config 4" "--config 6" "--shard-of 8" "--config 2"; do
  n=$(echo $c | tr -d ' -')
  ( timeout 300 python bench.py $c --no-cpu-baseline --no-other-configs --no-other-modes --no-streamed ) > $OUT/bench_$n.log 2>&1
  tail -1 $OUT/bench_$n.log | cut -c1-700
done
