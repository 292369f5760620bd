#!/bin/bash
# stream kernel: occupancy caps (7 / 8 waves per SIMD) and two tiles in flight, after the VALU trim
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s18
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python tools/sweep.py --shapes 8x4x16x200000000,8x4x16x50000000 --only stream_d4_u4_l4 --reps 7 --out $OUT/sweep.json ) > $OUT/sweep.log 2>&1; grep -v "^W\|^E\|amdgpu.ids" $OUT/sweep.log | tail -16
