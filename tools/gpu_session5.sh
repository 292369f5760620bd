#!/bin/bash
mkdir -p gpurun_out && cd "$GRAFT_REPO_ROOT"
export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out
( timeout 900 python -m pytest tests -m gpu -x -q ) > $OUT/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $OUT/pytest_gpu.log
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 2 --warmup 1 --rows 4000000 --backend gloo --chunk-rows 1000000 ) > $OUT/bench_2rank_gloo.log 2>&1; echo "bench2 rc=$?"
( timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 2 --warmup 1 --rows 4000000 --backend gloo --chunk-rows 1000000 --combine chain ) > $OUT/bench_2rank_gloo_chain.log 2>&1; echo "bench2chain rc=$?"
( timeout 900 python bench.py ) > $OUT/bench.log 2>&1; echo "bench rc=$?"
tail -4 $OUT/pytest_gpu.log; grep '^{' $OUT/bench_2rank_gloo.log | cut -c1-400; grep '^{' $OUT/bench_2rank_gloo_chain.log | cut -c1-300; tail -5 $OUT/bench_2rank_gloo.log | cut -c1-300; grep '^{' $OUT/bench.log
