#!/bin/bash
# Round 5, GPU session 1: the new tests (ADVICE fixes, deep kernels), config 6 on the deep kernel vs the generic kernel, the default bench line
# with other_configs, rocprofv3 stats of config 6.   Usage: tools/gpu_r05_s1.sh <tag>  -> gpurun_out/<tag>/
set -u
tag=${1:-r05_s1}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_q16_deep.py tests/test_q16_persistent.py -x -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/tests_new.log; tail -5 $OUT/tests_new.log
( timeout 300 python bench.py --config 6 --no-streamed ) > $OUT/bench_cfg6.log 2> $OUT/bench_cfg6.err; tail -1 $OUT/bench_cfg6.log | cut -c1-400
( timeout 300 python bench.py --config 6 --variant 0 --steps 2 --warmup 1 --no-cpu-baseline --no-streamed ) > $OUT/bench_cfg6_generic.log 2> $OUT/bench_cfg6_generic.err; tail -1 $OUT/bench_cfg6_generic.log | cut -c1-300
( timeout 300 python bench.py --config 6 --shard-of 8 --steps 3 --warmup 1 ) > $OUT/bench_cfg6_shard8.log 2>/dev/null; tail -1 $OUT/bench_cfg6_shard8.log | cut -c1-300
for d in 9 10 11 13 14; do
  ( timeout 300 python bench.py --config 6 --levels $d --trees 256 --rows 4000000 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed ) > $OUT/bench_d$d.log 2>/dev/null; tail -1 $OUT/bench_d$d.log | cut -c1-200
  ( timeout 300 python bench.py --config 6 --levels $d --trees 256 --rows 4000000 --variant 0 --steps 2 --warmup 1 --no-cpu-baseline --no-streamed ) > $OUT/bench_d${d}_generic.log 2>/dev/null; tail -1 $OUT/bench_d${d}_generic.log | cut -c1-200
done
( timeout 600 python bench.py ) > $OUT/bench_default.log 2> $OUT/bench_default.err; tail -1 $OUT/bench_default.log | cut -c1-300
B="python $GRAFT_REPO_ROOT/bench.py --config 6 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed"
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/stats_cfg6 -o bench -- $B ) > $OUT/stats_cfg6.log 2>&1; echo "cfg6 stats rc=$?"
( timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/gpu_tests.log; tail -3 $OUT/gpu_tests.log
