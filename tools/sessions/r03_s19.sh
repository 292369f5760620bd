#!/bin/bash
# stream kernel: rate against the number of rows (50 M rows ran at 5.6 TB/s, 200 M at 5.0 in session 18)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s19
rm -rf "$OUT"; mkdir -p "$OUT"
for n in 12500000 25000000 50000000 100000000 150000000 200000000 400000000; do
  ( timeout 300 python tools/run_shape.py --trees 8 --levels 4 --features 16 --rows $n --reps 7 --variant stream_d4_u4_l4_p2 ) 2>&1 | grep -v "^W\|^E\|amdgpu.ids" | tail -1
done | tee $OUT/rows_scan.log
