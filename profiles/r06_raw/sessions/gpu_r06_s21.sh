#!/bin/bash
# Round 6, session 21: the fixed part of a depth-8 scoring launch (tile load, prologue, epilogue): kernel time against the tree count at 50 M tuples.
set -u
tag=${1:-r06_s21}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
for T in 8 16 32 64 128 256 512; do
  rm -rf /tmp/prof_$T
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/prof_$T -o p -- python $GRAFT_REPO_ROOT/tools/run_shape.py --trees $T --levels 8 --features 32 --rows 50000000 --reps 3 ) > $OUT/run_$T.log 2>&1; grep "ms/launch" $OUT/run_$T.log | cut -c1-200 | tee -a $OUT/sweep.log; tail -3 $OUT/run_$T.log | cut -c1-300
  python tools/kstats.py /tmp/prof_$T | sed "s/^/[T=$T] /" | tee -a $OUT/sweep.log
done
