#!/bin/bash
# Round 6, session 18: sparse_r with the walk pipelined across PU groups (the next group's top levels behind every deep round): parity, then config 4 A/B.
set -u
tag=${1:-r06_s18}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_sparse_r.py tests/test_sparse.py tests/test_gpu_parity.py -m gpu -q -x ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
B="python bench.py --config 4 --no-cpu-baseline --no-other-configs --no-other-modes --no-streamed"
run() { name=$1; shift; ( timeout 300 env "$@" $B ) > $OUT/bench_$name.log 2>&1; python - "$OUT/bench_$name.log" "$name" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if not l: print(sys.argv[2], "NO LINE"); sys.exit()
d=json.loads(l[-1]); r=d['roofline']
print(sys.argv[2], d['value'], 'ms', d['ms_per_step'], 'kernel', r['kernel'], r['kernel_ms'], 'prepass', r['prepass_ms'], d.get('parity'))
PY
}
run pl1 DDT_SPARSE_R_PL=1
run pl0 DDT_SPARSE_R_PL=0
run pl1b DDT_SPARSE_R_PL=1
run pl0b DDT_SPARSE_R_PL=0
