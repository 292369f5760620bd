#!/bin/bash
# Round 6, session 4: rank32 with feature -> XCD affinity (each XCD's L2 holds W / 8 features' key blocks).
set -u
tag=${1:-r06_s4}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 600 python -m pytest tests/test_sparse.py -m gpu -x -q -k "every_sparse_kernel_variant" ) > $OUT/pytest_variants.log 2>&1; tail -3 $OUT/pytest_variants.log
B="python bench.py --config 4 --no-cpu-baseline --no-other-configs --no-other-modes --no-streamed"
run() { name=$1; shift; ( timeout 300 env "$@" $B ) > $OUT/bench_$name.log 2>&1; python - "$OUT/bench_$name.log" "$name" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
if not l: print(sys.argv[2], "NO LINE"); sys.exit()
d=json.loads(l[-1]); r=d['roofline']
print(sys.argv[2], d['value'], 'ms', d['ms_per_step'], 'kernel', r['kernel'], r['kernel_ms'], 'prepass', r['prepass_ms'], d.get('parity'))
PY
}
run default X=1
run sc1 DDT_R32_POLICY=1
run blk3 DDT_R32_BLK_LOG2=3
