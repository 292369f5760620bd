#!/bin/bash
# Round 6, session 8: the driver's default command with the per-rank proxies on its line; smoke; the tests touched since session 7.
set -u
tag=${1:-r06_s8}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 300 python -c "import __graft_entry__ as g; g.smoke()" ) > $OUT/smoke.log 2>&1; tail -2 $OUT/smoke.log
( timeout 1200 python -m pytest tests/test_sparse_r.py tests/test_q16_deep.py tests/test_zz_late_gpu.py tests/test_multiclass.py -m gpu -q -x --durations=5 ) > $OUT/pytest.log 2>&1; tail -12 $OUT/pytest.log
( time timeout 600 python bench.py ) > $OUT/bench_default.log 2>&1; tail -5 $OUT/bench_default.log | cut -c1-300
python - <<'PY'
import json
l=[x for x in open("gpurun_out/r06_s8/bench_default.log") if x.startswith('{')]
d=json.loads(l[-1])
print(json.dumps(d["other_modes"]["per_rank_proxies"], indent=1)[:3000])
print({k:(v.get("value"), v.get("kernel")) for k,v in d["other_configs"].items() if isinstance(v, dict)})
PY
