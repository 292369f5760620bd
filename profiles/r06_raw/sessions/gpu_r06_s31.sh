#!/bin/bash
# s31: slices finer than the clusters (a partial sum per PU group): GPU tests, latency against the batch size for the automatic rule, and forced slice counts
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_s31
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_q16_cluster_split.py tests/test_q16_persistent.py -q -x 2>&1 | grep -v "Extension modules" ) > $OUT/tests.log; tail -3 $OUT/tests.log
R=1,64,1024,4096,16384,32768,65536,131072,262144,524288,1048576,2097152
( timeout 600 python tools/latency_probe.py --configs 3 --rows $R --json $OUT/lat_cfg3_auto.json ) > $OUT/lat_cfg3_auto.log 2>&1; tail -12 $OUT/lat_cfg3_auto.log | cut -c60-175
for g in 0 16 32 64 125; do
  echo "groups=$g"
  ( timeout 600 python tools/latency_probe.py --configs 3 --rows 1024,4096,16384,32768,65536 --no-check --opt q16_split_groups=$g --json $OUT/lat_cfg3_g$g.json ) > $OUT/lat_cfg3_g$g.log 2>&1; tail -5 $OUT/lat_cfg3_g$g.log | cut -c60-175
done
( timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streamed --no-other-configs ) > $OUT/bench_cfg3.log 2>&1; tail -1 $OUT/bench_cfg3.log | cut -c1-200
