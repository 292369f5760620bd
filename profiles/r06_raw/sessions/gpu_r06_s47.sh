#!/bin/bash
# session r06_s47: randomized soak (tools/soak_fuzz.py, three seeds) + the GPU suite twice more at HEAD (flakiness check)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s47; mkdir -p $O
for s in 101 202 303; do
  timeout 600 python tools/soak_fuzz.py --seed $s --seconds 420 > $O/soak_$s.log 2>&1; echo "soak $s rc=$?"; tail -2 $O/soak_$s.log | cut -c1-600
done
for i in 1 2; do
  ( timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -3 ) > $O/gpu_tests_$i.log; tail -1 $O/gpu_tests_$i.log
done
