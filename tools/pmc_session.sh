#!/bin/bash
# rocprofv3 passes over one command (run on the GPU box): kernel-trace stats + one --pmc pass per counter group.
# Usage: tools/pmc_session.sh <tag> "<command>" "CTR1 CTR2" "CTR3" ...   -> gpurun_out/<tag>/{stats,pmcN}
set -u
tag=$1; cmd=$2; shift 2
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
mkdir -p "$OUT"
( cd /tmp && timeout 600 rocprofv3 --kernel-trace --stats -d "$OUT/stats" -o run -- $cmd ) > "$OUT/stats.log" 2>&1; echo "stats rc=$?"
i=0
for ctrs in "$@"; do
  i=$((i + 1))
  ( cd /tmp && timeout 600 rocprofv3 --pmc $ctrs -d "$OUT/pmc$i" -o pmc -- $cmd ) > "$OUT/pmc$i.log" 2>&1; echo "pmc$i ($ctrs) rc=$?"
done
tail -2 "$OUT/stats.log"
