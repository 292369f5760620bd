#!/bin/bash
# round 4, session 7: ensembles scored in parts (more than 32767 thresholds per feature) -- parity and the rate against the fp32 tile kernel
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s7
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 400 python -m pytest tests/test_q16.py tests/test_q16_persistent.py tests/test_comm_gpu.py -q -m gpu 2>&1 | tail -8 ) > $OUT/tests.log; grep -n "passed\|failed\|error" $OUT/tests.log | tail -3
V() { python - "$1" <<'PY'
import sys
sys.path.insert(0, "distributed-decisiontrees_amd")
import ddt
print(ddt.variant_names().index(sys.argv[1]))
PY
}
VT=$(V d8_t1024_r1_c4_u4_dma_f)
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-streamed --trees 4000 --features 16 --rows 20000000"
( DDT_DEBUG_PREPASS=1 timeout 120 $B ) > $OUT/t4000_parts.log 2>$OUT/t4000_parts.err; tail -1 $OUT/t4000_parts.log | cut -c1-240; grep "parts" $OUT/t4000_parts.err | head -2
( timeout 120 $B --variant $VT ) > $OUT/t4000_fp32tile.log 2>/dev/null; tail -1 $OUT/t4000_fp32tile.log | cut -c1-240
( timeout 120 python bench.py --steps 3 --warmup 1 --trees 4000 --features 16 --rows 20000000 --no-streamed ) > $OUT/t4000_parts_parity.log 2>/dev/null; python - $OUT/t4000_parts_parity.log <<'PY'
import json,sys
l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); print(l["value"], l["parity"], l["config"]["kernel"])
PY
