"""The reference's second multi-device mode (trees replicated, tuples partitioned, results interleaved:
rtl/DTEngine/DTInference.sv:28-37, PCIeReceiver.sv:289-312) -- SR.RowShardedScorer: no arithmetic crosses
devices, so the gathered scores are bit-identical to a single-engine run."""
import os
import socket

import numpy as np
import pytest

from oracle import oracle as O
import ddt
from tests import sharded_ref as SR

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        T, D, F, n = 600, 8, 32, 7001  # 600 trees per replica: the rank-quantised kernel is what runs
        w, f = ddt.synth_model(T, D, F)
        e = ddt.Engine(0)
        e.load_model(ddt.make_params(T, D, F), w, f)
        d = e.synth_tuples_device(0, n, F)
        got = SR.RowShardedScorer(e).score(d)
        torch.cuda.synchronize()
        want = O.score(O.Model(O.make_params(T, D, F), w, f), d.cpu().numpy().view(np.uint32))
        ret[rank] = bool(np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32))) and \
            e.info().variant_name.decode().startswith("q16")
        e.close()
    finally:
        dist.destroy_process_group()


def test_row_sharded_two_ranks_on_one_gpu():
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret.get(0) and ret.get(1), dict(ret)
