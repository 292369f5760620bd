#!/bin/bash
# s45: config 4 on ONE box, alternating libraries: the sparse_r object of the commit before the cut (gpurun_ab/libddt_old_sparse_r.so) against the new one
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_s45
rm -rf "$OUT"; mkdir -p "$OUT"
L=distributed-decisiontrees_amd/lib/libddt.so
for i in 1 2 3; do
  for w in old_sparse_r new; do
    cp gpurun_ab/libddt_$w.so $L
    ( timeout 600 python bench.py --config 4 --steps 5 --warmup 2 --no-cpu-baseline --no-streamed ) > $OUT/bench_${w}_$i.log 2>&1
    python - <<PY
import json
j=json.loads(open("$OUT/bench_${w}_$i.log").read().strip().splitlines()[-1]); print("$w", j["value"], j["ms_per_step"], j["roofline"]["kernel_ms"], j["roofline"]["prepass_ms"])
PY
  done
done
