#!/bin/bash
# s37: the cut launch on every depth (stream-order images: a partial sum per PU group) -- GPU tests, latency of 1000 x d6 / 500 x d7 / 2000 x d4 calls
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_s37
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_q16_cluster_split.py -q -x 2>&1 | grep -v "Extension modules" ) > $OUT/tests.log; tail -3 $OUT/tests.log
R=1,1024,4096,16384,65536,262144,1048576
for mode in "cut:--opt q16_cluster_split=-1" "uncut:--opt q16_cluster_split=0"; do
  name=${mode%%:*}; opt=${mode#*:}
  echo "== $name"
  ( timeout 600 python tools/latency_probe.py --configs 106,107,104,2,3 --rows $R $opt --json $OUT/lat_$name.json ) > $OUT/lat_$name.log 2>&1
  python - <<PY
import json
rs=json.load(open("$OUT/lat_$name.json"))
for c in (106,107,104,2,3):
    print(c, [r["kernel"] for r in rs if r["config"]==c][0], " ".join(f"{r['rows']}:{r['us_median']}" for r in rs if r["config"]==c), all(r["bit_exact"] in (True,None) for r in rs))
PY
done
( timeout 600 python bench.py --config 2 --steps 30 --warmup 5 --no-cpu-baseline --no-streamed ) > $OUT/bench_cfg2.log 2>&1; tail -1 $OUT/bench_cfg2.log | cut -c1-200
( timeout 600 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streamed --no-other-configs ) > $OUT/bench_cfg3.log 2>&1; tail -1 $OUT/bench_cfg3.log | cut -c1-200
