#!/bin/bash
# Round 5, session 8: the multi-GPU branch of bench.py (one-rank communicators) on configs 5, 6, 4 and the hybrid / rows / chain shardings: the new parity leg on every path.
set -u
tag=${1:-r05_s8}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
run() { name=$1; shift; ( timeout 600 python bench.py "$@" ) > $OUT/$name.log 2> $OUT/$name.err; python - "$OUT/$name.log" <<'PY'
import json,sys
try:
    j=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(sys.argv[1].split("/")[-1], j["value"], j["config"]["kernel"], j["config"]["parallelism"], {k:j.get("parity",{}).get(k) for k in ("rows_checked","bit_exact","within_tolerance","required","labels_that_differ","rows_that_differ_from_chain_oracle")}, (j.get("cpu_baseline") or {}).get("value"))
except Exception as ex:
    print(sys.argv[1], "FAILED", ex)
PY
}
run force_cfg6 --config 6 --force-collectives --steps 3 --warmup 1 --no-streamed --cpu-seconds 3
run force_cfg6_chain --config 6 --force-collectives --combine chain --steps 3 --warmup 1 --no-streamed --cpu-seconds 3 --no-other-modes
run force_cfg5 --config 5 --force-collectives --steps 3 --warmup 1 --no-streamed --cpu-seconds 3
run force_cfg5_chain --config 5 --force-collectives --combine chain --steps 3 --warmup 1 --no-streamed --cpu-seconds 3
run force_cfg4 --config 4 --force-collectives --steps 3 --warmup 1 --no-streamed --cpu-seconds 3
run force_cfg2_hybrid --config 2 --force-collectives --shard hybrid --tree-ranks 1 --steps 5 --warmup 2 --no-streamed --cpu-seconds 2 --no-other-modes
run force_cfg3_hybrid_nogather --force-collectives --shard hybrid --tree-ranks 1 --no-gather --steps 2 --warmup 1 --no-streamed --cpu-seconds 2 --no-other-modes
run gloo2_cfg6 --config 6 --gpus 2 --backend gloo --rows 2000000 --steps 2 --warmup 1 --no-other-modes --cpu-seconds 1
run gloo2_rows --gpus 2 --backend gloo --shard rows --rows 2000000 --steps 2 --warmup 1 --no-other-modes --cpu-seconds 1
tail -3 $OUT/*.err | grep -i "error\|traceback" | head
