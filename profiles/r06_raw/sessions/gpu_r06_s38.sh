#!/bin/bash
# s38: the cut launch on the deep kernels (incl. ensembles in parts): GPU tests, latency of config 6 calls, config 6 / headline re-checked
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_s38
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_q16_cluster_split.py tests/test_q16_deep.py -q -x 2>&1 | grep -v "Extension modules" ) > $OUT/tests.log; tail -3 $OUT/tests.log
R=1,1024,4096,16384,65536,262144,1048576
for mode in "cut:--opt q16_cluster_split=-1" "uncut:--opt q16_cluster_split=0"; do
  name=${mode%%:*}; opt=${mode#*:}
  echo "== $name"
  ( timeout 600 python tools/latency_probe.py --configs 6 --rows $R $opt --json $OUT/lat_$name.json ) > $OUT/lat_$name.log 2>&1
  python - <<PY
import json
rs=json.load(open("$OUT/lat_$name.json"))
print(rs[0]["kernel"], " ".join(f"{r['rows']}:{r['us_median']}" for r in rs), all(r["bit_exact"] in (True,None) for r in rs))
PY
done
( timeout 600 python bench.py --config 6 --steps 5 --warmup 2 --no-cpu-baseline --no-streamed ) > $OUT/bench_cfg6.log 2>&1; tail -1 $OUT/bench_cfg6.log | cut -c1-200
