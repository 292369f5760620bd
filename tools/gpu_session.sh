#!/bin/bash
# One GPU-box session (run through gpurun): GPU tests, then whatever extra commands are passed as arguments, each
# logged under gpurun_out/<tag>/.  Usage: tools/gpu_session.sh <tag> [--no-tests] 'cmd1' 'cmd2' ...
set -u
tag=$1; shift
out=gpurun_out/$tag
mkdir -p "$out"
if [ "${1:-}" != "--no-tests" ]; then
  timeout 1200 python -m pytest tests -m gpu -x -q --durations=8 > "$out/gpu_tests.log" 2>&1
  echo "gpu tests rc=$?" | tee -a "$out/gpu_tests.log"
  tail -5 "$out/gpu_tests.log"
else
  shift
fi
i=0
for cmd in "$@"; do
  i=$((i + 1))
  echo "=== [$i] $cmd" | tee "$out/cmd$i.log"
  timeout 600 bash -c "$cmd" >> "$out/cmd$i.log" 2>&1
  echo "rc=$?" >> "$out/cmd$i.log"
  tail -25 "$out/cmd$i.log"
done
