#!/bin/bash
# s43: the LDS-resident pre-passes with a unit per ticket on batches of up to 64 tiles (DDT_PREPASS_SMALL=0: four units as before): tests, per-call latency A/B
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_s43
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1200 python -m pytest tests/test_rank_transform.py tests/test_q16_cluster_split.py tests/test_q16.py -q -x -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/tests.log; tail -3 $OUT/tests.log
for small in 1 0; do
  echo "== DDT_PREPASS_SMALL=$small"
  ( DDT_PREPASS_SMALL=$small timeout 600 python tools/latency_probe.py --configs 3,2,106,6,5 --rows 1,1024,4096,16384,65536 --json $OUT/lat_small$small.json ) > $OUT/lat_small$small.log 2>&1
  python - <<PY
import json
rs=json.load(open("$OUT/lat_small$small.json"))
for c in (3,2,106,6,5):
    print(c, " ".join(f"{r['rows']}:{r['us_median']}" for r in rs if r["config"]==c), all(r["bit_exact"] in (True,None) for r in rs if r["config"]==c))
PY
done
