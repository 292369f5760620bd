#!/bin/bash
# round 4, session 5: HEAD check -- full GPU suite, the default bench line (with the sum_mode 2 leg and roofline.step), configs 1/2/4/5, what masked CUs cost a shard
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s5
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 500 python -m pytest tests -q -m gpu 2>&1 | grep -v "Extension modules" | tail -15 ) > $OUT/gpu_tests.log; grep -n "passed\|failed\|error" $OUT/gpu_tests.log | tail -3
( timeout 120 python bench.py ) > $OUT/bench_cfg3.log 2> $OUT/bench_cfg3.err; tail -1 $OUT/bench_cfg3.log | cut -c1-400
for cfg in 1 2 4 5; do
  ( timeout 120 python bench.py --config $cfg --no-cpu-baseline --no-streamed ) > $OUT/bench_cfg$cfg.log 2> $OUT/bench_cfg$cfg.err; tail -1 $OUT/bench_cfg$cfg.log | cut -c1-250
done
( timeout 120 python tools/cu_mask_probe.py ) > $OUT/cu_mask_shard8.json 2> $OUT/cu_mask_shard8.err; cat $OUT/cu_mask_shard8.json | cut -c1-900
( timeout 120 python tools/cu_mask_probe.py --shard-of 2 --masked 0,16,32 ) > $OUT/cu_mask_shard2.json 2> $OUT/cu_mask_shard2.err; cat $OUT/cu_mask_shard2.json | cut -c1-600
( timeout 60 python bench.py --steps 3 --warmup 1 --shard-of 8 --no-cpu-baseline --no-streamed ) > $OUT/bench_shard_of_8.log 2>/dev/null; tail -1 $OUT/bench_shard_of_8.log | cut -c1-300
( timeout 90 python bench.py --steps 3 --warmup 1 --force-collectives --shard hybrid --tree-ranks 1 --no-cpu-baseline --no-streamed ) > $OUT/bench_force_hybrid.log 2>$OUT/bench_force_hybrid.err; tail -1 $OUT/bench_force_hybrid.log | cut -c1-300
