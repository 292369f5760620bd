#!/bin/bash
# does a third resident block per CU hide the tile switch?  16 features: the rank tile is 32 KiB, three 1024-thread blocks fit a CU (plain launch).
# slope / fixed term of the depth-8 scoring kernel at F = 16 against F = 32 (trees 125, 250, 1000; 100 M tuples)
set -u
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/r04_s25; rm -rf "$OUT"; mkdir -p "$OUT"
for F in 16 32; do
  for T in 128 256 1000; do
    ( timeout 300 python bench.py --features $F --trees $T --steps 5 --warmup 2 --no-cpu-baseline --no-streamed --no-other-modes ) > $OUT/f${F}_t$T.log 2> $OUT/f${F}_t$T.err
    tail -1 $OUT/f${F}_t$T.log | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d.get('roofline') or {}
print('F=$F T=$T', d['ms_per_step'], r.get('kernel'), r.get('kernel_ms'), r.get('prepass_ms'))"
  done
done
