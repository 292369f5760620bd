#!/bin/bash
# round 4, session 8: config 5 with the classes' all-padding sub-group skipped; the persistent kernel's tests
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s8
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 400 python -m pytest tests/test_q16_persistent.py tests/test_multiclass.py tests/test_q16.py -q -m gpu 2>&1 | tail -8 ) > $OUT/tests.log; grep -n "passed\|failed\|error" $OUT/tests.log | tail -3
B="python bench.py --steps 10 --warmup 3 --no-streamed --config 5"
( timeout 120 $B ) > $OUT/cfg5.log 2>/dev/null; python - $OUT/cfg5.log <<'PY'
import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(l["value"], l["ms_per_step"], l["config"]["kernel"], l.get("parity"))
PY
( timeout 120 $B --no-cpu-baseline ) > $OUT/cfg5_b.log 2>/dev/null; tail -1 $OUT/cfg5_b.log | cut -c1-200
