#!/bin/bash
# s44: the sparse_r launch cut into slices of C groups: tests, config 4 throughput (the uncut path's prologue / epilogue changed), per-call latency
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_s44
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1200 python -m pytest tests/test_sparse_r.py -q -x -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/tests.log; tail -3 $OUT/tests.log
for i in 1 2; do ( timeout 600 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-streamed ) > $OUT/bench_cfg4_$i.log 2>&1; python - <<PY
import json
j=json.loads(open("$OUT/bench_cfg4_$i.log").read().strip().splitlines()[-1]); print("cfg4", j["value"], j["ms_per_step"], j["roofline"]["kernel"], j["roofline"]["kernel_ms"], j["roofline"]["prepass_ms"])
PY
done
R=1,256,1024,4096,16384,65536,262144,1048576
for mode in "cut:--opt q16_cluster_split=-1 --opt sparse_split_max_tiles=100000" "uncut:--opt q16_cluster_split=0"; do
  name=${mode%%:*}; opt=${mode#*:}
  echo "== $name"
  ( timeout 600 python tools/latency_probe.py --configs 4 --rows $R $opt --json $OUT/lat_$name.json ) > $OUT/lat_$name.log 2>&1 || tail -5 $OUT/lat_$name.log
  python - <<PY
import json
rs=json.load(open("$OUT/lat_$name.json"))
print(rs[0]["kernel"], " ".join(f"{r['rows']}:{r['us_median']}" for r in rs), all(r["bit_exact"] in (True,None) for r in rs))
PY
done
