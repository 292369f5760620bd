#!/bin/bash
# s30: the cluster split of small batches -- GPU tests, per-call latency against the batch size with the split off / forced / automatic, and the
# headline re-checked on the same box
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_s30
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_q16_cluster_split.py tests/test_q16_persistent.py -q -x 2>&1 | grep -v "Extension modules" ) > $OUT/tests.log; tail -3 $OUT/tests.log
R=1,64,1024,4096,16384,65536,131072,262144,524288,1048576,2097152,4194304
( timeout 600 python tools/latency_probe.py --configs 3 --rows $R --opt q16_cluster_split=0 --json $OUT/lat_cfg3_off.json ) > $OUT/lat_cfg3_off.log 2>&1; tail -12 $OUT/lat_cfg3_off.log
( timeout 600 python tools/latency_probe.py --configs 3 --rows $R --opt q16_cluster_split=1 --no-check --json $OUT/lat_cfg3_forced.json ) > $OUT/lat_cfg3_forced.log 2>&1; tail -12 $OUT/lat_cfg3_forced.log
( timeout 900 python tools/latency_probe.py --configs 3,2,6,4 --json $OUT/lat_auto.json ) > $OUT/lat_auto.log 2>&1; tail -32 $OUT/lat_auto.log
( timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streamed --no-other-configs ) > $OUT/bench_cfg3.log 2>&1; tail -1 $OUT/bench_cfg3.log | cut -c1-200
