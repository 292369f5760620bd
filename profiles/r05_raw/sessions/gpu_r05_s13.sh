#!/bin/bash
# Round 5, session 13 (needs lib/libddt_old.so = a build of the commit before, made by hand): same-box A/B of two BUILDS on config 4 (lib/libddt_old.so = the commit before, lib/libddt.so = the working tree)
set -u
tag=${1:-r05_s13}
cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$tag
rm -rf "$OUT"; mkdir -p "$OUT"
L=distributed-decisiontrees_amd/lib
cp $L/libddt.so $L/libddt_new.so
for rep in 1 2; do
  for which in old new; do
    cp $L/libddt_$which.so $L/libddt.so
    ( timeout 300 python tools/sparse_sweep.py --rows 4000000 --reps 3 --only sparse_dm1_k8_u8_t256,sparse_dk_k8_u8_t256 --out $OUT/sweep_${which}_$rep.json ) 2>&1 | grep variant | cut -c1-60,200-330 | sed "s/^/$which /"
  done
done
cp $L/libddt_new.so $L/libddt.so
