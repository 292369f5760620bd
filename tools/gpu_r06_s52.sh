#!/bin/bash
# session r06_s52 (at HEAD, after the evidence run): three more soak seeds + the GPU suite twice more (flakiness check)
set -u
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06_s52; mkdir -p $O
for s in 606 707 808; do
  timeout 600 python tools/soak_fuzz.py --seed $s --seconds 400 > $O/soak_$s.log 2>&1; echo "soak $s rc=$?"; tail -1 $O/soak_$s.log | cut -c1-160
done
for i in 1 2; do
  ( timeout 900 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | grep -v "Extension modules" ) > $O/gpu_tests_$i.log; grep -n "passed\|failed" $O/gpu_tests_$i.log | tail -1
done
