#!/bin/bash
# pre-pass / scoring overlap (option prepass_overlap_rows): A/B on config 3, the 125-tree shard of an 8-way job and config 2, then the
# GPU tests of the overlapped path and of the N>1 bench line's roofline object
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s40
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 170 python tools/overlap_ab.py --out $OUT/overlap_ab.json ) 2>&1 | grep -v "^W\|amdgpu.ids" | cut -c1-330 | tee $OUT/overlap_ab.log
( timeout 150 python -m pytest tests/test_q16.py::test_prepass_scoring_overlap "tests/test_zz_late_gpu.py::test_multi_gpu_branch_in_a_one_rank_communicator" -m gpu -x -q 2>&1 | tail -12 ) > $OUT/tests.log; cat $OUT/tests.log
