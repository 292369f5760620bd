#!/bin/bash
# config 1: phased result stores with the automatic choice (5 blocks x 12 slots), its GPU tests, the bench line
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s16; rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 600 python -m pytest tests/test_stream_phased.py -q -x -m gpu 2>&1 | tail -15 ) > $OUT/test_phased.log; tail -5 $OUT/test_phased.log
( timeout 300 python tools/stream_phase_ab.py --reps 5 --grid 0:1:0,0:0:0,0:1:0,0:0:0,6:0:0,4:0:0 ) > $OUT/ab_default.json 2> $OUT/ab_default.err; cat $OUT/ab_default.json | cut -c1-330
( timeout 600 python bench.py --config 1 --no-cpu-baseline --no-streamed ) > $OUT/bench_cfg1.log 2> $OUT/bench_cfg1.err; tail -1 $OUT/bench_cfg1.log | cut -c1-600
( timeout 600 python -m pytest tests -q -m gpu 2>&1 | tail -3 ) > $OUT/parity.log; cat $OUT/parity.log
