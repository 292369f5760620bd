#!/bin/bash
# the wide-tuple sparse fallback kernel on the GPU: its own test, every sparse variant forced by id (it is one of them), the classes path, smoke
set -u
cd "$GRAFT_REPO_ROOT"
OUT=$GRAFT_REPO_ROOT/gpurun_out/r03_s46
rm -rf "$OUT"; mkdir -p "$OUT"
( timeout 38 python -m pytest tests/test_sparse.py::test_gpu_sparse_tuples_too_wide_for_a_feature_tile tests/test_sparse.py::test_gpu_every_sparse_kernel_variant tests/test_sparse.py::test_gpu_sparse_classes tests/test_sparse.py::test_gpu_sparse_tree_shards_and_chain tests/test_sparse.py::test_gpu_sparse_single_leaf_trees_and_clusters tests/test_sparse.py::test_gpu_sparse_rejects_what_the_format_forbids -m gpu -x -q 2>&1 | tail -15 ) > $OUT/tests.log; cat $OUT/tests.log
