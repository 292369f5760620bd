#!/bin/bash
# round 4, session 9: where the persistent kernel's per-tile time goes -- static tile assignment (no ticket atomic) / no flag load, on the 125-tree shard
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s9
rm -rf "$OUT"; mkdir -p "$OUT"
B="python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-streamed --shard-of 8"
run() { name=$1; shift; ( timeout 90 $B "$@" ) > $OUT/$name.log 2>$OUT/$name.err; python - "$OUT/$name.log" "$name" <<'PY'
import json,sys
try:
    l=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=l.get("roofline") or {}
    print(sys.argv[2], l["ms_per_step"], "ms; kernel", r.get("kernel"), r.get("kernel_ms"), "pre", r.get("prepass_ms"))
except Exception as ex:
    print(sys.argv[2], "FAILED", ex)
PY
}
run x
run p --opt q16_persistent=1
run p_static --opt q16_persistent=1 --opt q16_prepass_nt=4
run p_noflag --opt q16_persistent=1 --opt q16_prepass_nt=8
run p_both --opt q16_persistent=1 --opt q16_prepass_nt=12
run x2
