#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s3
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 300 tools/ubench/ubench walk ) > $OUT/ubench_walk.json 2> $OUT/ubench_walk.err; echo "ubench walk rc=$?"; cat $OUT/ubench_walk.json
