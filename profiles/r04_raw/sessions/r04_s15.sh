#!/bin/bash
# config 1: phased result stores, tuning grid (blocks per CU x slots x window)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s15; rm -rf "$OUT"; mkdir -p "$OUT"
G="0:1:0"
for b in 4 5; do for nb in 6 8 12 16 21; do for w in 1500 2000 2500 3000 4000; do G="$G,$b:$nb:$w"; done; done; done
G="$G,6:7:1500,6:7:2000,6:7:2500,3:24:3000,3:24:4000,0:1:0"
( timeout 600 python tools/stream_phase_ab.py --reps 4 --grid "$G" ) > $OUT/grid_cfg1.json 2> $OUT/grid_cfg1.err; tail -2 $OUT/grid_cfg1.err
python - <<'P'
import json
for l in open("gpurun_out/r04_s15/grid_cfg1.json"):
    d=json.loads(l); print(d["blocks_per_cu"],d["res_tiles"],d["window_ticks"],d["ms"],d["gtuples_per_s"],d["equals_direct_stores"])
P
