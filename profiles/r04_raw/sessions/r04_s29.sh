#!/bin/bash
# sum_mode 2: one v_max3_u16 per two sums + one test per fold instead of a test per sum -- same-box A/B against the previous build, every test that runs sum_mode 2
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s29; rm -rf "$OUT"; mkdir -p "$OUT"
LIB=distributed-decisiontrees_amd/lib/libddt.so
cp $LIB /tmp/libddt_head.so
line() { python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('$1', d['value'], d['ms_per_step'], r.get('kernel'), r.get('kernel_ms'))"; }
for lib in head prev head prev; do
  if [ $lib = head ]; then cp /tmp/libddt_head.so $LIB; else cp gpurun_ab/libddt_prev.so $LIB; fi
  ( timeout 300 python bench.py --steps 5 --warmup 2 --sum-mode 2 --no-cpu-baseline --no-streamed --no-other-modes ) > $OUT/full_${lib}_sum2.log 2> $OUT/full_${lib}_sum2.err
  tail -1 $OUT/full_${lib}_sum2.log | line "$lib full sum_mode=2"
done
cp /tmp/libddt_head.so $LIB
( timeout 300 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streamed --no-other-modes ) > $OUT/full_head_sum0.log 2>/dev/null; tail -1 $OUT/full_head_sum0.log | line "head full sum_mode=0"
( timeout 300 python bench.py --config 2 --sum-mode 2 --no-cpu-baseline --no-streamed --no-other-modes ) > $OUT/cfg2_head_sum2.log 2>/dev/null; tail -1 $OUT/cfg2_head_sum2.log | line "head cfg2 sum_mode=2"
( timeout 900 python -m pytest tests/test_adder_corner.py tests/test_gpu_parity.py tests/test_q16.py tests/test_q16_persistent.py tests/test_q16_padding_skip.py tests/test_stream_phased.py tests/test_full_size_gpu.py tests/test_sparse.py tests/test_multiclass.py tests/test_fuzz_gpu.py -q -x -m gpu 2>&1 | grep -v "Extension modules" | tail -4 ) > $OUT/tests.log; tail -3 $OUT/tests.log
