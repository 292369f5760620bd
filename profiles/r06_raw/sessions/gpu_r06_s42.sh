#!/bin/bash
# s42: kernels of a 1024-row call per config (rocprofv3 --kernel-trace --stats; 40 calls each), cut automatically
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_s42
rm -rf "$OUT"; mkdir -p "$OUT"
for cfg in 2 106 5 6 4; do
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$cfg -o kt -- python $GRAFT_REPO_ROOT/tools/latency_probe.py --configs $cfg --rows 1024 --no-check --reps 40 ) > $OUT/kt_$cfg.log 2>&1
  echo "== config $cfg"; python tools/kstats.py $OUT/kt_$cfg | sed 's/ ms/ ms/' 
done
