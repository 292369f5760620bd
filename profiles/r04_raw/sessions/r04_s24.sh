#!/bin/bash
# where does a tile's time go in the persistent kernel?  (DDT_Q16P_PROFILE: phase sums by wave 0's 100 MHz clock) -- 125-tree shard, 1000 trees
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s24; rm -rf "$OUT"; mkdir -p "$OUT"
for sh in 8 0; do
  extra=""; [ $sh != 0 ] && extra="--shard-of $sh"
  ( DDT_Q16P_PROFILE=1 timeout 300 python bench.py $extra --steps 3 --warmup 1 --no-cpu-baseline --no-streamed --no-other-modes --opt q16_persistent=1 ) > $OUT/prof_shard$sh.log 2> $OUT/prof_shard$sh.err
  grep "q16p_" $OUT/prof_shard$sh.err | tail -2
  ( timeout 300 python bench.py $extra --steps 5 --warmup 2 --no-cpu-baseline --no-streamed --no-other-modes --opt q16_persistent=1 ) > $OUT/plain_shard$sh.log 2> $OUT/plain_shard$sh.err
  tail -1 $OUT/plain_shard$sh.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('shard-of $sh', d['ms_per_step'], r.get('kernel'), r.get('kernel_ms'), r.get('prepass_ms'))"
done
