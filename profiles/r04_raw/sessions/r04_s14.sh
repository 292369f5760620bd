#!/bin/bash
# config 1: the stream kernel with phased result stores, A/B against its direct stores (full-output comparison), ragged sizes
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s14; rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 300 python tools/stream_phase_ab.py ) > $OUT/ab_cfg1.json 2> $OUT/ab_cfg1.err; cat $OUT/ab_cfg1.json | cut -c1-330; tail -3 $OUT/ab_cfg1.err
( timeout 200 python tools/stream_phase_ab.py --rows 20000003 --reps 2 --grid 0:0:0,5:0:200,6:3:100,2:0:1000 ) > $OUT/ab_ragged.json 2>&1; cat $OUT/ab_ragged.json | cut -c1-330
( timeout 200 python tools/stream_phase_ab.py --rows 1000 --reps 1 --grid 0:0:0,5:0:200 ) > $OUT/ab_small.json 2>&1; cat $OUT/ab_small.json | cut -c1-330
( timeout 200 python tools/stream_phase_ab.py --trees 24 --levels 6 --features 28 --rows 50000000 --reps 3 --grid 0:1:0,0:0:0 ) > $OUT/ab_t24.json 2>&1; cat $OUT/ab_t24.json | cut -c1-330
( timeout 300 python -m pytest tests/test_gpu_parity.py -q -x -m gpu 2>&1 | tail -3 ) > $OUT/parity.log; cat $OUT/parity.log
