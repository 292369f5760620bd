#!/bin/bash
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s12
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python bench.py --steps 3 --warmup 1 --no-cpu-baseline ) > $OUT/bench_streamed.log 2> $OUT/bench_streamed.err; python - <<PY
import json
j=json.loads(open("$OUT/bench_streamed.log").read().strip().splitlines()[-1])
print(j["value"], j["ms_per_step"], json.dumps(j.get("streamed"), indent=1))
PY
( timeout 600 python tools/feeder_bench.py 16000000 ) 2>&1 | grep -v "^W\|amdgpu" | tail -14
