#!/bin/bash
# s32: where a small call's 65 us go: kernel timeline of 1024-row calls (cut into PU groups / clusters / uncut), config 2 for comparison
set -u
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r06_s32
rm -rf "$OUT"; mkdir -p "$OUT"
for mode in "auto:--opt q16_split_groups=-1:3" "clusters:--opt q16_split_groups=0:3" "uncut:--opt q16_cluster_split=0:2"; do
  name=${mode%%:*}; rest=${mode#*:}; opt=${rest%:*}; per=${rest##*:}
  ( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_$name -o kt -- python $GRAFT_REPO_ROOT/tools/latency_probe.py --configs 3 --rows 1024 --no-check --reps 40 $opt ) > $OUT/kt_$name.log 2>&1
  echo "== $name"; python tools/ktimeline.py $OUT/kt_$name --per-call $per --calls 30
done
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/kt_cfg2 -o kt -- python $GRAFT_REPO_ROOT/tools/latency_probe.py --configs 2 --rows 1024 --no-check --reps 40 ) > $OUT/kt_cfg2.log 2>&1
echo "== cfg2"; python tools/ktimeline.py $OUT/kt_cfg2 --per-call 2 --calls 30
