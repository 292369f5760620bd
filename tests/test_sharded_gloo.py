"""World-size-2 (and 4) gloo process groups on CPU driving the PYTHON MIRROR of the multi-GPU pipeline (tests/sharded_ref.py, test
infrastructure) with per-rank partial scores supplied by the oracle: what this covers is the shard arithmetic, the chunk schedule and the
chain order as a specification -- NOT the product's C++ pipeline (csrc/ddt_comm.cpp), which has no CPU backend to talk to.  The product
pipeline's multi-rank coverage is tests/test_comm_mock.py / test_engine_mock.py (the C++ compiled unchanged against a deferred-execution
model of HIP streams + RCCL, 2-8 ranks, incl. the hybrid jobs) and tests/test_comm_gpu.py (one-rank communicators on the real RCCL)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import oracle as O
import ddt
from tests import sharded_ref as SR


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, mode, chunk_rows, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        T, D, F, n = 100, 6, 28, 1003
        m = O.gen_model(T, D, F, dist=1)
        x = O.gen_tuples(5, n, F, dist=1)
        b, e = ddt.shard_bounds(T, world)[rank]

        def partial(tuples, out):
            out.copy_(torch.from_numpy(O.score_shard(m, tuples.numpy().view(np.uint32), b, e)))

        sc = SR.ShardedScorer(partial, ddt.tuple_words(F), mode=mode, chunk_rows=chunk_rows)
        got = sc.score(torch.from_numpy(x.view(np.int32))).numpy()
        want_chain = O.score(m, x, n_devices=world)
        gold = O.score(m, x, want_gold=True)[1]
        if mode == "chain" or world == 2:  # two-term fp32 add is order independent
            ok = np.array_equal(got.view(np.uint32), want_chain.view(np.uint32))
        else:
            ok = np.allclose(got, gold, rtol=1e-6, atol=1e-6)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,chunk", [(2, "allreduce", 400), (2, "chain", 400), (2, "chain", 1 << 20),
                                              (4, "chain", 257), (4, "allreduce", 1 << 20)])
def test_sharded_scorer_gloo(world, mode, chunk):
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, mode, chunk, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)


def test_chain_sum_is_the_reference_hop_order():
    parts = torch.tensor([[1.0], [2.0 ** -24], [2.0 ** -24], [2.0 ** -24]], dtype=torch.float32)
    # ((1 + e) + e) + e = 1 (each add ties to even); any pairwise order would give 1 + 2^-23
    assert SR.chain_sum(parts)[0].item() == 1.0


def _cls_worker(rank, world, port, mode, chunk_rows, ret):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        T, D, F, K, n = 96, 5, 12, 4, 777
        m = O.gen_model(T, D, F, dist=1, clusters=2)
        x = O.gen_tuples(8, n, F, dist=1)
        per = T // K  # class-major layout: class k = trees [k*per, (k+1)*per); this rank holds shard `rank` of each class

        def partial(tuples, out):
            t = tuples.numpy().view(np.uint32)
            for k in range(K):
                b, e = ddt.shard_bounds(per, world)[rank]
                out[k].copy_(torch.from_numpy(O.score_shard(m, t, k * per + b, k * per + e)))

        def argmax(scores):
            return torch.from_numpy(np.argmax(scores.numpy(), axis=0).astype(np.int32))  # first maximum wins

        sc = SR.ShardedClassifier(partial, ddt.tuple_words(F), K, argmax, mode=mode, chunk_rows=chunk_rows)
        labels, scores = sc.classify(torch.from_numpy(x.view(np.int32)))
        want_l, want_cs = O.classify(m, x, K, interleaved=False, n_devices=world)
        if mode == "chain" or world == 2:
            ok = np.array_equal(scores.numpy().view(np.uint32), want_cs.view(np.uint32)) and np.array_equal(labels.numpy(), want_l)
        else:
            ok = np.allclose(scores.numpy(), want_cs, rtol=1e-6, atol=1e-6)
        ret[rank] = bool(ok)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,mode,chunk", [(2, "allreduce", 300), (2, "chain", 300), (4, "chain", 1 << 20), (4, "allreduce", 200)])
def test_sharded_classifier_gloo(world, mode, chunk):
    """config 5 across ranks: per-class partial sums combined (all-reduce / deterministic chain), then argmax."""
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    port = _free_port()
    procs = [ctx.Process(target=_cls_worker, args=(r, world, port, mode, chunk, ret)) for r in range(world)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(180)
        assert p.exitcode == 0
    assert all(ret.get(r) for r in range(world)), dict(ret)
