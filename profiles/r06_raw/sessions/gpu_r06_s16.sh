#!/bin/bash
# Round 6, session 16: the LDS-resident rank pre-passes after the VALU diet (byte-address probes with DS immediates, linear tables, IEEE as a template parameter).
set -u
tag=${1:-r06_s16}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 1500 python -m pytest tests/test_q16.py tests/test_q16_persistent.py tests/test_q16_padding_skip.py tests/test_gpu_parity.py tests/test_multiclass.py tests/test_fuzz_gpu.py tests/test_sparse.py tests/test_full_size_gpu.py -m gpu -q -x ) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log
for c in "--config 3" "--shard-of 8" "--config 2" "--config 5" "--config 6"; do
  n=$(echo $c | tr -d ' -')
  ( timeout 300 python bench.py $c --no-cpu-baseline --no-other-configs --no-other-modes --no-streamed ) > $OUT/bench_$n.log 2>&1
  python - $OUT/bench_$n.log "$c" <<'PY'
import json,sys
l=[x for x in open(sys.argv[1]) if x.startswith('{')]
d=json.loads(l[-1]); r=d['roofline']
print(sys.argv[2], d['value'], 'ms', d['ms_per_step'], 'kernel', r['kernel'], r['kernel_ms'], 'prepass', r['prepass_ms'])
PY
done
