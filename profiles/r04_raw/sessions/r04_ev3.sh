#!/bin/bash
# closing call of round 4 at HEAD (rebuilt after the comment-only reference updates): GPU suite, smoke, default bench line, rocprofv3 stats of
# configs 2 and 5, the shard line on 10 steps
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_ev3; rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1500 python -m pytest tests -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/gpu_tests.log; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -2
( timeout 300 python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1; grep "smoke ok" $OUT/smoke.log
( timeout 900 python bench.py ) > $OUT/bench_cfg3.log 2> $OUT/bench_cfg3.err; tail -1 $OUT/bench_cfg3.log | cut -c1-260
for cfg in 2 5; do
  B="python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 10 --warmup 3 --no-cpu-baseline --no-streamed"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_cfg$cfg -o bench -- $B ) > $OUT/stats_cfg$cfg.log 2>&1; echo "cfg$cfg stats rc=$?"
done
( timeout 600 python bench.py --steps 10 --warmup 3 --shard-of 8 --no-cpu-baseline --no-streamed ) > $OUT/bench_shard_of_8.log 2>/dev/null; tail -1 $OUT/bench_shard_of_8.log | cut -c1-200
