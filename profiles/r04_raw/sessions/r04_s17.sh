#!/bin/bash
# config 1: wave-private tiles (no block barrier per tile) on top of the phased result stores
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s17; rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 600 python -m pytest tests/test_stream_phased.py tests/test_gpu_parity.py tests/test_fuzz_gpu.py -q -x -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/tests.log; grep -n "passed\|failed\|rror" $OUT/tests.log | tail -5
( timeout 300 python tools/stream_phase_ab.py --reps 5 --grid 0:1:0,0:0:0,0:1:0,0:0:0,6:0:0,4:0:0,5:0:2000,5:0:4000 ) > $OUT/ab_default.json 2> $OUT/ab_default.err; cat $OUT/ab_default.json | cut -c1-200
( timeout 600 python bench.py --config 1 --no-cpu-baseline --no-streamed ) > $OUT/bench_cfg1.log 2> $OUT/bench_cfg1.err; tail -1 $OUT/bench_cfg1.log | cut -c1-200
