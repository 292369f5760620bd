#!/bin/bash
# s36: the whole GPU suite + the driver's command at the commit that cut small batches into slices
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_s36
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1800 python -m pytest tests -q -m gpu 2>&1 | grep -v "Extension modules" ) > $OUT/gpu_tests.log; tail -3 $OUT/gpu_tests.log
( timeout 900 python bench.py ) > $OUT/bench_cfg3.log 2> $OUT/bench_cfg3.err; python - <<PY
import json
j=json.loads(open("$OUT/bench_cfg3.log").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], json.dumps(j["other_modes"]["small_batches"]))
print({k:v.get("value") for k,v in j["other_configs"].items()})
PY
