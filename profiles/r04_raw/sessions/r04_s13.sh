#!/bin/bash
# config 1, third look, second pass: window length, buffer size, blocks per CU, a barrier per tile: does the TIME at which result stores reach memory matter?  (tools/ubench/storephase.hip)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s13; rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 200 tools/ubench/storephase ) > $OUT/storephase.json 2> $OUT/storephase.err; cat $OUT/storephase.json | cut -c1-260; tail -3 $OUT/storephase.err
