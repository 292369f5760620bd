"""One-rank RCCL sanity on a single GPU: the calls bench.py / the C++ pipeline (csrc/ddt_comm.cpp) and its Python mirror tests/sharded_ref.py make at N > 1 (init with device_id, async
all_reduce + wait, all_to_all_single, all_gather_into_tensor, barrier, MAX reduce of a float64)."""
import os
import sys

import torch
import torch.distributed as dist

os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29555")
os.environ.setdefault("RANK", "0")
os.environ.setdefault("WORLD_SIZE", "1")
torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
x = torch.arange(1 << 20, dtype=torch.float32, device="cuda")
w = dist.all_reduce(x[: 1 << 19], op=dist.ReduceOp.SUM, async_op=True)
w.wait()
a = torch.empty(1 << 10, device="cuda")
dist.all_to_all_single(a, x[: 1 << 10].contiguous())
g = torch.empty(1 << 10, device="cuda")
dist.all_gather_into_tensor(g, a)
t = torch.tensor([1.5], dtype=torch.float64, device="cuda")
dist.all_reduce(t, op=dist.ReduceOp.MAX)
dist.barrier()
torch.cuda.synchronize()
assert torch.equal(g, x[: 1 << 10]) and t.item() == 1.5
print("nccl one-rank sanity ok", torch.__version__, dist.get_backend())
dist.destroy_process_group()
