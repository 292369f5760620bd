#!/bin/bash
# Round 6, session 22: sparse_r with wave-private rows (one-instruction feature address): parity suite, config 4, the shape sweep.
set -u
tag=${1:-r06_s22}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 900 python -m pytest tests/test_sparse_r.py tests/test_sparse.py tests/test_zz_late_gpu.py -m gpu -x -q ) > $OUT/tests.log 2>&1; tail -3 $OUT/tests.log
for i in 1 2; do ( timeout 600 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-streamed ) 2>/dev/null | tail -1 | python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']
print('cfg4', d['value'], d['ms_per_step'], r['kernel'], r['kernel_ms'], r['prepass_ms'], d.get('parity'))" | tee -a $OUT/cfg4.log; done
shape() { for o in 1 0; do ( timeout 300 python tools/run_shape.py --sparse --rows ${ROWS:-4000000} --reps 3 --opt sparse_r32=$o "$@" ) 2>&1 | tail -1 | cut -c1-220 | sed "s/^/[r32=$o $*] /"; done; }
shape --trees 512 --levels 16 --features 64 2>&1 | tee -a $OUT/sweep.log
shape --trees 512 --levels 16 --features 64 --bins 255 2>&1 | tee -a $OUT/sweep.log
shape --trees 256 --levels 15 --features 48 2>&1 | tee -a $OUT/sweep.log
shape --trees 256 --levels 15 --features 40 --full-levels 8 2>&1 | tee -a $OUT/sweep.log
shape --trees 256 --levels 12 --features 64 2>&1 | tee -a $OUT/sweep.log
shape --trees 128 --levels 14 --features 20 --full-levels 6 2>&1 | tee -a $OUT/sweep.log
shape --trees 512 --levels 10 --features 64 --full-levels 8 2>&1 | tee -a $OUT/sweep.log
shape --trees 1000 --levels 13 --features 28 --full-levels 7 --permille 750 2>&1 | tee -a $OUT/sweep.log
