#!/bin/bash
# "_xh": the pinned depth-8 kernel on half tiles, three 512-thread blocks per CU -- shard-of-8 / shard-of-4 / full against "_x" (variant 6 vs 7), parity tests
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s26; rm -rf "$OUT"; mkdir -p "$OUT"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$1', d['value'], d['ms_per_step'], r.get('kernel'), r.get('kernel_ms'), r.get('prepass_ms'))"; }
for v in 6 7 6 7; do
  ( timeout 300 python bench.py --shard-of 8 --variant $v --steps 10 --warmup 3 --no-cpu-baseline --no-streamed --no-other-modes ) > $OUT/shard8_v$v.log 2> $OUT/shard8_v$v.err
  tail -1 $OUT/shard8_v$v.log | line "shard-of-8 variant $v"
done
for v in 6 7; do
  ( timeout 300 python bench.py --shard-of 4 --variant $v --steps 5 --warmup 2 --no-cpu-baseline --no-streamed --no-other-modes ) > $OUT/shard4_v$v.log 2> $OUT/shard4_v$v.err
  tail -1 $OUT/shard4_v$v.log | line "shard-of-4 variant $v"
  ( timeout 300 python bench.py --variant $v --steps 5 --warmup 2 --no-cpu-baseline --no-streamed --no-other-modes ) > $OUT/full_v$v.log 2> $OUT/full_v$v.err
  tail -1 $OUT/full_v$v.log | line "full variant $v"
done
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_q16.py tests/test_adder_corner.py -q -x -m gpu 2>&1 | grep -v "Extension modules" | tail -5 ) > $OUT/tests.log; tail -3 $OUT/tests.log
