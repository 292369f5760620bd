#!/bin/bash
# s33: the LDS-staged combine: GPU tests, kernel timeline of a one-tile call, latency against the batch size (automatic / clusters only / uncut)
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_s33
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_q16_cluster_split.py -q -x 2>&1 | grep -v "Extension modules" ) > $OUT/tests.log; tail -3 $OUT/tests.log
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_auto -o kt -- python $GRAFT_REPO_ROOT/tools/latency_probe.py --configs 3 --rows 1024 --no-check --reps 40 ) > $OUT/kt_auto.log 2>&1
python tools/ktimeline.py $OUT/kt_auto --per-call 3 --calls 30
R=1,1024,2048,4096,8192,16384,32768,65536,131072,262144,524288,1048576,2097152
for mode in "auto:--opt q16_split_groups=-1" "clusters:--opt q16_split_groups=0" "uncut:--opt q16_cluster_split=0"; do
  name=${mode%%:*}; opt=${mode#*:}
  echo "== $name"
  ( timeout 600 python tools/latency_probe.py --configs 3 --rows $R --no-check $opt --json $OUT/lat_$name.json ) > $OUT/lat_$name.log 2>&1
  python - <<PY
import json
print(" ".join(f"{r['rows']}:{r['us_median']}" for r in json.load(open("$OUT/lat_$name.json"))))
PY
done
