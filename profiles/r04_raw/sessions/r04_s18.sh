#!/bin/bash
# (a) does phasing the writes pay for the rank pre-pass's 2:1 read:write pattern?  (tools/ubench/prepassphase.hip)
# (b) config 1 at HEAD: launch-to-launch spread of the phased stream kernel, bench line with 30 steps
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s18; rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 240 tools/ubench/prepassphase ) > $OUT/prepassphase.json 2> $OUT/prepassphase.err; cut -c1-330 $OUT/prepassphase.json; tail -3 $OUT/prepassphase.err
( timeout 200 python tools/stream_phase_ab.py --reps 20 --grid 0:1:0,0:0:0,0:1:0,0:0:0 ) > $OUT/ab_spread.json 2> $OUT/ab_spread.err; cut -c1-330 $OUT/ab_spread.json
( timeout 300 python bench.py --config 1 --steps 30 --warmup 5 --no-cpu-baseline --no-streamed ) > $OUT/bench_cfg1.log 2> $OUT/bench_cfg1.err; tail -1 $OUT/bench_cfg1.log | cut -c1-300
( timeout 300 python -m pytest tests/test_stream_phased.py -q -x -m gpu 2>&1 | tail -3 ) > $OUT/test_phased.log; cat $OUT/test_phased.log
