#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s7
mkdir -p "$OUT"
( timeout 900 python tools/sweep.py --shapes 100x6x28x10000000,200x6x28x10000000,50x8x32x10000000,100x8x32x10000000,30x6x16x10000000 --only d6_t1024,d8_t1024_r1_c4_u4_dma_f,q16_d6,q16_d8_c8_u4_gl --reps 5 --out $OUT/sweep_fr.json ) > $OUT/sweep.log 2>&1; grep -v "^W\|^E\|amdgpu.ids" $OUT/sweep.log | tail -30
