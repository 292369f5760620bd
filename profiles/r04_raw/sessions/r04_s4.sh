#!/bin/bash
# round 4, session 4: the persistent kernel with its prefetch behind the first chunk barrier; full GPU suite; sum_mode 2 at HEAD
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s4
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 500 python -m pytest tests/test_q16_persistent.py tests/test_multiclass.py tests/test_comm_gpu.py tests/test_gpu_parity.py -q -m gpu 2>&1 | grep -v "Extension modules" | tail -15 ) > $OUT/gpu_tests.log; grep -n "passed\|failed\|error" $OUT/gpu_tests.log | tail -3
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streamed"
V() { python - "$1" <<'PY'
import sys
sys.path.insert(0, "distributed-decisiontrees_amd")
import ddt
print(ddt.variant_names().index(sys.argv[1]))
PY
}
VX=$(V q16_d8_c8_u4_gl_s2_cm_x); VP=$(V q16_d8_c8_u4_gl_s2_cm_p)
run() { name=$1; shift; ( timeout 90 $B "$@" ) > $OUT/$name.log 2>$OUT/$name.err; python - "$OUT/$name.log" "$name" <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=l.get("roofline") or {}
    print(sys.argv[2], l["value"], "Mtuples/s", l["ms_per_step"], "ms; kernel", r.get("kernel"), r.get("kernel_ms"), "pre", r.get("prepass_ms"))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
run shard8_x --shard-of 8 --variant $VX
run shard8_p --shard-of 8 --variant $VP
run shard4_x --shard-of 4 --variant $VX
run shard4_p --shard-of 4 --variant $VP


run cfg3_x --variant $VX
run cfg3_p --variant $VP
run cfg3_x_sum2 --variant $VX --sum-mode 2
run cfg3_base
run cfg3_p_sum2 --variant $VP --sum-mode 2
run cfg5_x --config 5 --variant $VX
run cfg5_p --config 5 --variant $VP


