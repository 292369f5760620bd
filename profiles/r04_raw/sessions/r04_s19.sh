#!/bin/bash
# plain q16 kernels, depth <= 6: the EMPTY padding trees of the last chunk are not walked -- GPU tests, config 2 A/B
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s19; rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_q16_padding_skip.py tests/test_gpu_parity.py tests/test_fuzz_gpu.py -q -x -m gpu 2>&1 | grep -v "Extension modules" | tail -8 ) > $OUT/tests.log; tail -4 $OUT/tests.log
for w in 1 0 1 0; do
  ( timeout 300 python bench.py --config 2 --steps 20 --warmup 5 --no-cpu-baseline --no-streamed --opt q16_walk_padding=$w ) > $OUT/bench_cfg2_walk$w.log 2> $OUT/bench_cfg2_walk$w.err
  tail -1 $OUT/bench_cfg2_walk$w.log | python -c "import sys,json; d=json.loads(sys.stdin.read()); r=d['roofline']; print('walk_padding=$w', d['value'], d['ms_per_step'], r['kernel'], r['kernel_ms'], r['prepass_ms'])"
done
