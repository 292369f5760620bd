#!/bin/bash
mkdir -p gpurun_out
export HSA_ENABLE_IPC_MODE_LEGACY=0
for n in 2 4 8; do
  timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port $((29500+n)) bench.py --gpus $n --backend gloo --rows 4000000 --chunk-rows 1000000 --steps 2 --warmup 1 > gpurun_out/s23_n$n.log 2>&1
  tail -1 gpurun_out/s23_n$n.log | cut -c1-330
done
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29600 bench.py --gpus 2 --backend gloo --combine chain --rows 4000000 --chunk-rows 1000000 --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-330
timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29601 bench.py --gpus 2 --backend gloo --shard rows --rows 4000000 --steps 2 --warmup 1 2>&1 | tail -1 | cut -c1-330
timeout 600 python bench.py --rows 20000000 --steps 2 --warmup 1 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['cpu_baseline'], d['parity'])"
