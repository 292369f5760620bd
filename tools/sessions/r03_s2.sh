#!/bin/bash
# round 3, GPU session 2: microbenchmarks again (walk forms fixed against loop-invariant hoisting, visit-sequence encodings)
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s2
rm -rf "$OUT"; mkdir -p "$OUT"
for what in valu walk; do
  ( timeout 300 tools/ubench/ubench $what ) > $OUT/ubench_$what.json 2> $OUT/ubench_$what.err; echo "ubench $what rc=$?"; cat $OUT/ubench_$what.json
done
