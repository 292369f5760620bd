#!/bin/bash
# one-launch multi-class calls: the last round's tiles split by class (mc_tail_split) -- GPU tests, config 5 A/B in one run
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s27; rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_q16_persistent.py tests/test_multiclass.py "tests/test_full_size_gpu.py::test_config5_every_label_and_class_sum" -q -x -m gpu 2>&1 | grep -v "Extension modules" | tail -8 ) > $OUT/tests.log; tail -4 $OUT/tests.log
for sp in 0 -1 0 -1; do
  ( timeout 300 python bench.py --config 5 --steps 10 --warmup 3 --no-cpu-baseline --no-streamed --no-other-modes --opt mc_tail_split=$sp ) > $OUT/cfg5_split$sp.log 2> $OUT/cfg5_split$sp.err
  tail -1 $OUT/cfg5_split$sp.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('mc_tail_split=$sp', d['value'], d['ms_per_step'], (d.get('checked') or {}))"
done
