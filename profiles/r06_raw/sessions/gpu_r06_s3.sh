#!/bin/bash
# Round 6, session 3: sparse_r after the half-round walk; what bounds the rank32 pre-pass (cache policy of its gathers, block size); kernel split by rocprofv3.
set -u
tag=${1:-r06_s3}
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
run nt DDT_R32_POLICY=2
run blk3 DDT_R32_BLK_LOG2=3
run blk4 DDT_R32_BLK_LOG2=4
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/prof -o cfg4 -- python $GRAFT_REPO_ROOT/bench.py --config 4 --steps 3 --warmup 1 --no-cpu-baseline --no-other-configs --no-other-modes --no-streamed ) > $OUT/rocprof.log 2>&1
python tools/prof_summary.py r06_s3_cfg4 --stats $OUT/prof --kernel score_sparse_r > $OUT/prof_summary.log 2>&1 || true; cp profiles/r06_s3_cfg4.md $OUT/ 2>/dev/null
head -20 $OUT/r06_s3_cfg4.md
