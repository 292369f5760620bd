#!/bin/bash
# round 4, session 6: what unevenly / evenly masked CUs cost the plain (_x) and the persistent (_p, dynamic tile tickets) kernel; pinned order at depth 6
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s6
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_q16.py -q -m gpu 2>&1 | tail -8 ) > $OUT/tests.log; grep -n "passed\|failed\|error" $OUT/tests.log | tail -3
V() { python - "$1" <<'PY'
import sys
sys.path.insert(0, "distributed-decisiontrees_amd")
import ddt
print(ddt.variant_names().index(sys.argv[1]))
PY
}
VX=$(V q16_d8_c8_u4_gl_s2_cm_x); VP=$(V q16_d8_c8_u4_gl_s2_cm_p); V6=$(V q16_d6_c16_u4_s2)
for v in x p; do
  vid=$VX; [ $v = p ] && vid=$VP
  ( timeout 120 python tools/cu_mask_probe.py --variant $vid --masked 0,16,32,64 ) > $OUT/cu_mask_even_$v.json 2>/dev/null; cut -c1-700 $OUT/cu_mask_even_$v.json
  ( timeout 120 python tools/cu_mask_probe.py --variant $vid --masked 0,8,16 --one-xcd ) > $OUT/cu_mask_onexcd_$v.json 2>/dev/null; cut -c1-600 $OUT/cu_mask_onexcd_$v.json
done
B="python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-streamed"
( timeout 60 $B --config 2 ) > $OUT/cfg2_auto.log 2>/dev/null; tail -1 $OUT/cfg2_auto.log | cut -c1-200
( timeout 60 $B --config 2 --variant $V6 ) > $OUT/cfg2_s2.log 2>/dev/null; tail -1 $OUT/cfg2_s2.log | cut -c1-200
( timeout 60 $B --config 2 ) > $OUT/cfg2_auto2.log 2>/dev/null; tail -1 $OUT/cfg2_auto2.log | cut -c1-200
python - $OUT <<'PY'
import json,sys,glob
for f in sorted(glob.glob(sys.argv[1]+"/cfg2_*.log")):
    l=json.loads(open(f).read().strip().splitlines()[-1]); r=l["roofline"]
    print(f.split("/")[-1], l["value"], l["ms_per_step"], r["kernel"], r["kernel_ms"], r["prepass_ms"])
PY
