#!/bin/bash
# hot copy of the chunk loop for sum_mode 2 (the reference adder): same-box A/B of HEAD against the previous build (gpurun_ab/libddt_prev.so),
# sum modes 0 and 2, headline + config 5; then the parity tests that run sum_mode 2 through every variant
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s22; rm -rf "$OUT"; mkdir -p "$OUT"
LIB=distributed-decisiontrees_amd/lib/libddt.so
cp $LIB /tmp/libddt_head.so
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$1', d['value'], d['ms_per_step'], r.get('kernel'), r.get('kernel_ms'), r.get('prepass_ms'))"; }
for lib in head prev head prev; do
  if [ $lib = head ]; then cp /tmp/libddt_head.so $LIB; else cp gpurun_ab/libddt_prev.so $LIB; fi
  for sm in 0 2; do
    ( timeout 300 python bench.py --steps 5 --warmup 2 --sum-mode $sm --no-cpu-baseline --no-streamed --no-other-modes ) > $OUT/full_${lib}_sum$sm.log 2> $OUT/full_${lib}_sum$sm.err
    tail -1 $OUT/full_${lib}_sum$sm.log | line "$lib full sum_mode=$sm"
  done
  ( timeout 300 python bench.py --config 5 --steps 10 --warmup 3 --sum-mode 2 --no-cpu-baseline --no-streamed --no-other-modes ) > $OUT/cfg5_${lib}_sum2.log 2> $OUT/cfg5_${lib}_sum2.err
  tail -1 $OUT/cfg5_${lib}_sum2.log | line "$lib cfg5 sum_mode=2"
done
cp /tmp/libddt_head.so $LIB
( timeout 900 python -m pytest tests/test_adder_corner.py tests/test_gpu_parity.py tests/test_q16_persistent.py tests/test_q16.py -q -x -m gpu 2>&1 | grep -v "Extension modules" | tail -5 ) > $OUT/tests.log; tail -3 $OUT/tests.log
