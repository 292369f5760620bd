#!/bin/bash
# start-up stagger of a CU's two blocks (option q16_stagger): 125-tree shard (plain _x and persistent _p), the full 1000 trees, config 5;
# and the padding-skip GPU tests that s19 stopped in front of
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s20; rm -rf "$OUT"; mkdir -p "$OUT"
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$1', d['value'], d['ms_per_step'], r.get('kernel'), r.get('kernel_ms'), r.get('prepass_ms'), d.get('checked') or d.get('parity') or '')"; }
for st in 0 -1 1500 3000 6000 0 -1; do
  ( timeout 300 python bench.py --shard-of 8 --steps 10 --warmup 3 --no-cpu-baseline --no-streamed --no-other-modes --opt q16_stagger=$st ) > $OUT/shard_x_st$st.log 2> $OUT/shard_x_st$st.err
  tail -1 $OUT/shard_x_st$st.log | line "shard _x stagger=$st"
done
for st in 0 -1 3000 0 -1; do
  ( timeout 300 python bench.py --shard-of 8 --steps 10 --warmup 3 --no-cpu-baseline --no-streamed --no-other-modes --opt q16_persistent=1 --opt q16_stagger=$st ) > $OUT/shard_p_st$st.log 2> $OUT/shard_p_st$st.err
  tail -1 $OUT/shard_p_st$st.log | line "shard _p stagger=$st"
done
for st in 0 -1; do
  ( timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streamed --no-other-modes --opt q16_stagger=$st ) > $OUT/full_st$st.log 2> $OUT/full_st$st.err
  tail -1 $OUT/full_st$st.log | line "full stagger=$st"
  ( timeout 300 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-streamed --no-other-modes --opt q16_stagger=$st ) > $OUT/cfg5_st$st.log 2> $OUT/cfg5_st$st.err
  tail -1 $OUT/cfg5_st$st.log | line "cfg5 stagger=$st"
  ( timeout 300 python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-streamed --no-other-modes --opt q16_stagger=$st ) > $OUT/cfg2_st$st.log 2> $OUT/cfg2_st$st.err
  tail -1 $OUT/cfg2_st$st.log | line "cfg2 stagger=$st"
done
( timeout 900 python -m pytest tests/test_q16_padding_skip.py -q -x -m gpu 2>&1 | grep -v "Extension modules" | tail -8 ) > $OUT/tests.log; tail -3 $OUT/tests.log
