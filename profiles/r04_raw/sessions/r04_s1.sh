#!/bin/bash
# round 4, session 1: the persistent rank-quantised kernel (_p) and the one-launch classes -- parity first, then same-box A/B
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s1
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 300 python -m pytest tests/test_q16_persistent.py -x -q -m gpu 2>&1 | tail -15 ) > $OUT/tests_persistent.log; tail -4 $OUT/tests_persistent.log
( timeout 400 python -m pytest tests -q -m gpu 2>&1 | grep -v "Extension modules" | tail -15 ) > $OUT/gpu_tests.log; grep -n "passed\|failed\|error" $OUT/gpu_tests.log | tail -3
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streamed"
run() { name=$1; shift; ( timeout 90 $B "$@" ) > $OUT/$name.log 2>$OUT/$name.err; python - "$OUT/$name.log" "$name" <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=l.get("roofline") or {}
    print(sys.argv[2], l["value"], "Mtuples/s", l["ms_per_step"], "ms; kernel", r.get("kernel"), r.get("kernel_ms"), "pre", r.get("prepass_ms"), "parity", (l.get("parity") or {}).get("bit_exact"))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
run shard8_base --shard-of 8
run shard8_p --shard-of 8 --opt q16_persistent=1
run shard8_p_nt1 --shard-of 8 --opt q16_persistent=1 --opt q16_prepass_nt=1
run shard8_p_nt2 --shard-of 8 --opt q16_persistent=1 --opt q16_prepass_nt=2
run shard8_p_nt3 --shard-of 8 --opt q16_persistent=1 --opt q16_prepass_nt=3
run shard8_base2 --shard-of 8
run shard4_base --shard-of 4
run shard4_p --shard-of 4 --opt q16_persistent=1
run shard2_base --shard-of 2
run shard2_p --shard-of 2 --opt q16_persistent=1
run cfg3_base
run cfg3_p --opt q16_persistent=1
run cfg3_nt1 --opt q16_prepass_nt=1
run cfg5_base --config 5
run cfg5_p --config 5 --opt q16_persistent=1
run cfg5_base2 --config 5
