#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s5
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_adder_corner.py tests/test_q16.py -m gpu -x -q 2>&1 | tail -15 ) > $OUT/tests.log; cat $OUT/tests.log
for sm in 0 2; do
  ( timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-streamed --sum-mode $sm ) > $OUT/bench_sm$sm.log 2> $OUT/bench_sm$sm.err; tail -1 $OUT/bench_sm$sm.log | cut -c1-330
done
( timeout 600 python bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed --sum-mode 0 ) > $OUT/bench_c2_sm0.log 2>/dev/null; tail -1 $OUT/bench_c2_sm0.log | cut -c1-250
( timeout 600 python bench.py --config 2 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed --sum-mode 2 ) > $OUT/bench_c2_sm2.log 2>/dev/null; tail -1 $OUT/bench_c2_sm2.log | cut -c1-250
