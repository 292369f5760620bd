#!/bin/bash
# Round 5, GPU session 3: depth 15 / 16 deep kernels, the rank kernel's grid A/B, config 6 after the pre-pass changes, the full suite at HEAD.
set -u
tag=${1:-r05_s3}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_q16_deep.py -x -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/tests_deep.log; tail -3 $OUT/tests_deep.log
( timeout 300 python bench.py --config 6 --no-streamed ) > $OUT/bench_cfg6.log 2> $OUT/bench_cfg6.err; tail -1 $OUT/bench_cfg6.log | cut -c1-300
( DDT_RANK_GRID_OLD=1 timeout 300 python bench.py --config 6 --no-streamed --no-cpu-baseline ) > $OUT/bench_cfg6_oldgrid.log 2>/dev/null; tail -1 $OUT/bench_cfg6_oldgrid.log | cut -c1-300
S="python tools/sweep.py --reps 3"
( timeout 300 $S --shapes 1000x8x64x20000000 --only q16w_d8_c8_u4_gl_s2_cm_x --out $OUT/wide_new.json ) > $OUT/wide_newgrid.log 2>&1; tail -1 $OUT/wide_newgrid.log
( DDT_RANK_GRID_OLD=1 timeout 300 $S --shapes 1000x8x64x20000000 --only q16w_d8_c8_u4_gl_s2_cm_x --out $OUT/wide_old.json ) > $OUT/wide_oldgrid.log 2>&1; tail -1 $OUT/wide_oldgrid.log
( timeout 300 $S --shapes 4000x8x16x20000000 --only q16_d8_c8_u4_gl_s2_cm_x --out $OUT/t4000_new.json ) > $OUT/t4000_newgrid.log 2>&1; tail -1 $OUT/t4000_newgrid.log
( DDT_RANK_GRID_OLD=1 timeout 300 $S --shapes 4000x8x16x20000000 --only q16_d8_c8_u4_gl_s2_cm_x --out $OUT/t4000_old.json ) > $OUT/t4000_oldgrid.log 2>&1; tail -1 $OUT/t4000_oldgrid.log
for d in 15 16; do
  ( timeout 300 python bench.py --config 6 --levels $d --trees 64 --rows 4000000 --steps 3 --warmup 1 --no-cpu-baseline --no-streamed ) > $OUT/bench_d$d.log 2>/dev/null; tail -1 $OUT/bench_d$d.log | cut -c1-200
  ( timeout 300 python bench.py --config 6 --levels $d --trees 64 --rows 4000000 --variant 0 --steps 2 --warmup 1 --no-cpu-baseline --no-streamed ) > $OUT/bench_d${d}_generic.log 2>/dev/null; tail -1 $OUT/bench_d${d}_generic.log | cut -c1-200
done
( timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/gpu_tests.log; tail -3 $OUT/gpu_tests.log
