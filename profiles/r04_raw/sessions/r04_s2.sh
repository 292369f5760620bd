#!/bin/bash
# round 4, session 2: the pinned LDS read order (four chains in flight) on the plain (_x) and the persistent (_p) kernel; _pu = persistent, compiler's order
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s2
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 400 python -m pytest tests/test_q16_persistent.py tests/test_gpu_parity.py tests/test_multiclass.py tests/test_q16.py -x -q -m gpu 2>&1 | tail -15 ) > $OUT/tests.log; tail -4 $OUT/tests.log
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streamed"
V() { python - "$1" <<'PY'
import sys
sys.path.insert(0, "distributed-decisiontrees_amd")
import ddt
print(ddt.variant_names().index(sys.argv[1]))
PY
}
VX=$(V q16_d8_c8_u4_gl_s2_cm_x); VP=$(V q16_d8_c8_u4_gl_s2_cm_p); VPU=$(V q16_d8_c8_u4_gl_s2_cm_pu); VCM=$(V q16_d8_c8_u4_gl_s2_cm)
run() { name=$1; shift; ( timeout 90 $B "$@" ) > $OUT/$name.log 2>$OUT/$name.err; python - "$OUT/$name.log" "$name" <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=l.get("roofline") or {}
    print(sys.argv[2], l["value"], "Mtuples/s", l["ms_per_step"], "ms; kernel", r.get("kernel"), r.get("kernel_ms"), "pre", r.get("prepass_ms"))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
run shard8_base --shard-of 8
run shard8_cm --shard-of 8 --variant $VCM
run shard8_x --shard-of 8 --variant $VX
run shard8_p --shard-of 8 --variant $VP
run shard8_pu --shard-of 8 --variant $VPU
run shard4_x --shard-of 4 --variant $VX
run shard4_p --shard-of 4 --variant $VP
run cfg3_base
run cfg3_x --variant $VX
run cfg3_p --variant $VP
run cfg3_base2
run cfg5_base --config 5
run cfg5_x --config 5 --variant $VX
run cfg5_p --config 5 --variant $VP
