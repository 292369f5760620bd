#!/bin/bash
# Round 6 evidence on the GPU box (through gpurun), one call at HEAD: GPU tests, smoke, the bench lines (config 3 = the driver's command with
# other_configs, 6 and 4 in full, 1 / 2 / 5 without the CPU and host-feeder legs), rocprofv3 kernel-trace stats + FETCH / WRITE passes of the
# config-3, config-6, config-4 and config-1 bench commands, the shard-regime and one-rank-communicator lines.
# Usage: tools/gpu_evidence_r06.sh <tag>    -> gpurun_out/<tag>/...   (then tools/refresh_profiles.sh <tag> r06)
set -u
tag=${1:-r06_ev}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/gpu_tests.log; grep -n "passed\|failed" $OUT/gpu_tests.log | tail -2
( timeout 300 python __graft_entry__.py smoke ) > $OUT/smoke.log 2>&1; grep "smoke ok" $OUT/smoke.log
( timeout 900 python bench.py ) > $OUT/bench_cfg3.log 2> $OUT/bench_cfg3.err; tail -1 $OUT/bench_cfg3.log | cut -c1-300
( timeout 900 python bench.py --config 6 ) > $OUT/bench_cfg6.log 2> $OUT/bench_cfg6.err; tail -1 $OUT/bench_cfg6.log | cut -c1-300
( timeout 900 python bench.py --config 4 ) > $OUT/bench_cfg4.log 2> $OUT/bench_cfg4.err; tail -1 $OUT/bench_cfg4.log | cut -c1-300
for cfg in 1 2 5; do
  ( timeout 600 python bench.py --config $cfg --no-cpu-baseline --no-streamed ) > $OUT/bench_cfg$cfg.log 2> $OUT/bench_cfg$cfg.err; tail -1 $OUT/bench_cfg$cfg.log | cut -c1-300
done
for cfg in 3 6 4 1 2 5; do
  B="python $GRAFT_REPO_ROOT/bench.py --config $cfg --steps 3 --warmup 1 --no-cpu-baseline --no-streamed --no-other-configs"
  [ $cfg = 1 ] && B="python $GRAFT_REPO_ROOT/bench.py --config 1 --steps 10 --warmup 3 --no-cpu-baseline --no-streamed"
  [ $cfg = 2 ] && B="python $GRAFT_REPO_ROOT/bench.py --config 2 --steps 10 --warmup 3 --no-cpu-baseline --no-streamed"
  ( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/stats_cfg$cfg -o bench -- $B ) > $OUT/stats_cfg$cfg.log 2>&1; echo "cfg$cfg stats rc=$?"
  ( cd /tmp && timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/fetch_cfg$cfg -o pmc -- $B ) > $OUT/fetch_cfg$cfg.log 2>&1; echo "cfg$cfg fetch rc=$?"
  ( cd /tmp && timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/write_cfg$cfg -o pmc -- $B ) > $OUT/write_cfg$cfg.log 2>&1; echo "cfg$cfg write rc=$?"
done
( timeout 600 python bench.py --steps 3 --warmup 1 --force-collectives --no-streamed ) > $OUT/bench_force_allreduce.log 2>/dev/null
( timeout 600 python bench.py --steps 3 --warmup 1 --force-collectives --combine chain --no-streamed --no-other-modes ) > $OUT/bench_force_chain.log 2>/dev/null
( timeout 600 python bench.py --steps 3 --warmup 1 --shard-of 8 --no-cpu-baseline --no-streamed ) > $OUT/bench_shard_of_8.log 2>/dev/null
( timeout 600 python bench.py --config 6 --steps 3 --warmup 1 --shard-of 8 ) > $OUT/bench_cfg6_shard_of_8.log 2>/dev/null
( timeout 600 python bench.py --steps 3 --warmup 1 --force-collectives --shard hybrid --tree-ranks 1 --no-streamed --no-other-modes ) > $OUT/bench_force_hybrid.log 2>/dev/null
tail -qn1 $OUT/bench_force_*.log $OUT/bench_shard_of_8.log $OUT/bench_cfg6_shard_of_8.log | cut -c1-160
du -sh $OUT | tail -1
