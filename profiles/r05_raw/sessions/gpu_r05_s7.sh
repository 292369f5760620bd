#!/bin/bash
# Round 5, session 7: where the deep kernels stop paying (few trees): generic vs q16d per tree count.
set -u
tag=${1:-r05_s7}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
S="python tools/sweep.py --reps 3"
( timeout 600 $S --shapes 1x12x32x10000000,2x12x32x10000000,4x12x32x10000000,8x12x32x10000000,16x12x32x10000000,32x12x32x10000000,64x12x32x10000000 --only generic,q16d_d12 --out $OUT/small_d12.json ) > $OUT/small_d12.log 2>&1; grep "ok=" $OUT/small_d12.log
( timeout 600 $S --shapes 2x9x32x10000000,8x9x32x10000000,16x9x32x10000000,32x9x32x10000000,8x10x16x10000000,16x14x32x4000000,8x15x32x4000000,16x15x32x4000000 --only generic,q16d_d --out $OUT/small_other.json ) > $OUT/small_other.log 2>&1; grep "ok=" $OUT/small_other.log
( timeout 600 $S --shapes 8x12x64x10000000,16x12x64x10000000,32x12x64x10000000,8x9x48x10000000,32x9x48x10000000 --only generic,q16dw_d --out $OUT/small_wide.json ) > $OUT/small_wide.log 2>&1; grep "ok=" $OUT/small_wide.log
