#!/bin/bash
# last call of the round: bench.py --shard-of 8 (what one of 8 ranks computes), the one-rank-communicator line with its per-rank roofline, smoke
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s44
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 40 python bench.py --steps 3 --warmup 1 --shard-of 8 ) > $OUT/bench_shard_of_8.log 2>/dev/null; tail -1 $OUT/bench_shard_of_8.log | cut -c1-300
( timeout 50 python -m pytest tests/test_zz_late_gpu.py::test_one_shard_of_a_tree_sharded_job_on_one_gpu -q -m gpu 2>&1 | tail -3 ) > $OUT/tests.log; cat $OUT/tests.log
( timeout 50 python bench.py --steps 3 --warmup 1 --force-collectives --no-cpu-baseline --no-streamed ) > $OUT/bench_force_allreduce.log 2>/dev/null; tail -1 $OUT/bench_force_allreduce.log | cut -c1-200
( timeout 40 python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1; grep "smoke ok" $OUT/smoke.log
