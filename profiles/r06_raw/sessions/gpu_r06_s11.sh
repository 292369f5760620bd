#!/bin/bash
# Round 6, session 11: sparse_r after the VALU trims (two-instruction feature address, no select on a leaf child's word): parity + config 4.
set -u
tag=${1:-r06_s11}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_sparse_r.py tests/test_sparse.py -m gpu -q -x ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for i in 1 2; do
( timeout 300 python bench.py --config 4 --no-cpu-baseline --no-other-configs --no-other-modes --no-streamed ) > $OUT/bench_cfg4_$i.log 2>&1
python - $OUT/bench_cfg4_$i.log <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']
print(d['value'], 'ms', d['ms_per_step'], 'kernel', r['kernel'], r['kernel_ms'], 'prepass', r['prepass_ms'], d.get('parity'))
PY
done
