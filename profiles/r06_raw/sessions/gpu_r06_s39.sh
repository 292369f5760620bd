#!/bin/bash
# s39: the cut launch over the classes of a one-vs-all model (config 5's kernel): GPU tests, classify latency, config 5 re-checked
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_s39
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_q16_cluster_split.py tests/test_q16_persistent.py tests/test_multiclass.py -q -x 2>&1 | grep -v "Extension modules" ) > $OUT/tests.log; tail -3 $OUT/tests.log
R=1,1024,4096,16384,65536,262144,1048576
for mode in "cut:--opt q16_cluster_split=-1" "uncut:--opt q16_cluster_split=0"; do
  name=${mode%%:*}; opt=${mode#*:}
  echo "== $name"
  ( timeout 600 python tools/latency_probe.py --configs 5 --rows $R $opt --json $OUT/lat_$name.json ) > $OUT/lat_$name.log 2>&1 || tail -5 $OUT/lat_$name.log
  python - <<PY
import json
rs=json.load(open("$OUT/lat_$name.json"))
print(rs[0]["kernel"], " ".join(f"{r['rows']}:{r['us_median']}" for r in rs))
PY
done
( timeout 600 python bench.py --config 5 --steps 5 --warmup 2 --no-cpu-baseline --no-streamed ) > $OUT/bench_cfg5.log 2>&1; tail -1 $OUT/bench_cfg5.log | cut -c1-200
