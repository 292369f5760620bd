"""BASELINE config 5: 10-class one-vs-all, 100 trees/class, depth 8, per-class sums then argmax.
An EXTENSION (the reference has one fp32 sum per tuple and no classes): the oracle defines it as K independent
reference-order ensembles + argmax (lowest index on ties); labels must match bit for bit."""
import numpy as np
import pytest

from oracle import oracle as O
import ddt
from tests import sharded_ref as SR
from tests import refimpl as R


def test_oracle_classify_equals_per_class_scoring():
    T, D, F, K, n = 40, 5, 12, 4, 300
    m = O.gen_model(T, D, F, dist=1, clusters=2)
    x = O.gen_tuples(0, n, F, dist=1)
    thr, fidx, mr, leaf = R.unpack_model(m.wlines, m.flines, T, D, O.wlpt(D), O.flpt(D))
    lb = R.traverse_all(thr, fidx, mr, leaf, x, m.params.missing_bits)
    for inter in (True, False):
        labels, cs = O.classify(m, x, K, interleaved=inter)
        for k in range(K):
            ids = [i for i in range(T) if (i % K if inter else i // (T // K)) == k]
            want = R.score_reference_order(lb[:, ids], 2)
            assert np.array_equal(cs[k].view(np.uint32), want.view(np.uint32))
        assert np.array_equal(labels, np.argmax(cs, axis=0).astype(np.int32))  # numpy argmax: first max wins
    l2, cs2 = O.classify(m, x, K, n_devices=2)
    assert np.allclose(cs2, O.classify(m, x, K)[1], rtol=0, atol=1e-6)


def test_argmax_tie_goes_to_lowest_class():
    # two identical classes => every tuple ties => label 0
    thr = np.full((2, 1), 0.5, np.float32)
    m = O.pack_model(thr, np.zeros((2, 1), np.int64), np.zeros((2, 1), np.uint8),
                     np.array([[1.0, 2.0], [1.0, 2.0]], np.float32), 4)
    x = O.tuples_from_float(np.array([[0.1, 0, 0, 0], [0.9, 0, 0, 0]], np.float32))
    labels, cs = O.classify(m, x, 2)
    assert list(labels) == [0, 0] and np.array_equal(cs[0], cs[1])


@pytest.mark.gpu
@pytest.mark.parametrize("T,D,F,K,n,inter", [(1000, 8, 32, 10, 1500, True), (1000, 8, 32, 10, 700, False),
                                              (60, 6, 28, 3, 2000, True), (24, 4, 16, 2, 999, True),
                                              (35, 7, 20, 5, 600, True)])
def test_gpu_classify_matches_oracle(T, D, F, K, n, inter):
    import torch

    dist = 1 if D != 8 else 0
    w, f = ddt.synth_model(T, D, F, dist)
    C = ddt.default_clusters((T + K - 1) // K)
    m = O.Model(O.make_params(T, D, F, clusters=C), w, f)
    x = O.gen_tuples(2, n, F, dist=dist)
    want_l, want_cs = O.classify(m, x, K, interleaved=inter)
    e = ddt.Engine(0)
    e.load_model_multiclass(ddt.make_params(T, D, F, clusters=C), w, f, K, inter)
    assert e.info().num_classes == K and e.info().local_trees == T
    labels, cs = e.classify(x, want_scores=True)                                   # host feeder path
    assert np.array_equal(cs.view(np.uint32), want_cs.view(np.uint32))
    assert np.array_equal(labels, want_l)
    dl, dcs = e.classify_device(torch.from_numpy(x.view(np.int32)).cuda())          # device path
    torch.cuda.synchronize()
    assert np.array_equal(dl.cpu().numpy(), want_l) and np.array_equal(dcs.cpu().numpy().view(np.uint32), want_cs.view(np.uint32))
    with pytest.raises(ddt.DDTError):
        e.score(x)  # a multi-class engine refuses the single-score call
    # tree-sharded x2: per-class partials from two engines, chain add per class, argmax == oracle's 2-device model
    if T // K >= 2:
        want2_l, want2_cs = O.classify(m, x, K, interleaved=inter, n_devices=2)
        parts = []
        d = torch.from_numpy(x.view(np.int32)).cuda()
        for g in range(2):
            eg = ddt.Engine(0)
            eg.load_model_multiclass(ddt.make_params(T, D, F, clusters=C), w, f, K, inter, g, 2)
            parts.append(eg.classify_device(d, want_labels=False)[1])
            torch.cuda.synchronize()
            eg.close()
        stacked = torch.stack(parts).reshape(2, K * n).contiguous()
        comb = e.chain_sum_device(stacked).reshape(K, n)
        lab = e.argmax_device(comb.contiguous())
        torch.cuda.synchronize()
        assert np.array_equal(comb.cpu().numpy().view(np.uint32), want2_cs.view(np.uint32))
        assert np.array_equal(lab.cpu().numpy(), want2_l)
    e.close()


@pytest.mark.gpu
def test_classes_share_one_rank_prepass():
    """Multi-class models big enough for the rank-quantised path build ONE set of rank tables over all classes and
    run the transpose + rank pre-pass once per batch; missing values and negatives (slow image) included."""
    import torch

    T, D, F, K, n = 400, 8, 32, 4, 3000
    w, f = ddt.synth_model(T, D, F, 1)
    C = ddt.default_clusters(T // K)
    m = O.Model(O.make_params(T, D, F, clusters=C), w, f)
    x = O.gen_tuples(4, n, F, dist=1)
    want_l, want_cs = O.classify(m, x, K, interleaved=True)
    e = ddt.Engine(0)
    e.load_model_multiclass(ddt.make_params(T, D, F, clusters=C), w, f, K, True)
    assert e.info().variant_name.decode().startswith("q16")     # 100 trees per class, 400 in total
    dl, dcs = e.classify_device(torch.from_numpy(x.view(np.int32)).cuda())
    torch.cuda.synchronize()
    assert np.array_equal(dcs.cpu().numpy().view(np.uint32), want_cs.view(np.uint32))
    assert np.array_equal(dl.cpu().numpy(), want_l)
    labels, cs = e.classify(x, want_scores=True)                 # host feeder: two workspaces in flight
    assert np.array_equal(cs.view(np.uint32), want_cs.view(np.uint32)) and np.array_equal(labels, want_l)
    e.close()


@pytest.mark.gpu
def test_config5_full_batch_properties():
    """BASELINE config 5 at 10 M rows: deterministic, batch-split invariant, labels and per-class sums bit-exact on a
    strided sample that the oracle re-scores."""
    import torch

    T, D, F, K, N = 1000, 8, 32, 10, 10_000_000
    w, f = ddt.synth_model(T, D, F)
    C = ddt.default_clusters(T // K)
    e = ddt.Engine(0)
    e.load_model_multiclass(ddt.make_params(T, D, F, clusters=C), w, f, K, True)
    d = e.synth_tuples_device(0, N, F)
    l1, s1 = e.classify_device(d)
    l2, s2 = e.classify_device(d)
    torch.cuda.synchronize()
    assert torch.equal(l1, l2) and torch.equal(s1.view(torch.int32), s2.view(torch.int32))
    cut = 3_333_337
    la, sa = e.classify_device(d[:cut])
    lb, sb = e.classify_device(d[cut:])
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([la, lb]), l1)
    assert torch.equal(torch.cat([sa, sb], dim=1).view(torch.int32), s1.view(torch.int32))
    idx = torch.arange(0, N, 4999, device="cuda")
    xs = d[idx].cpu().numpy().view(np.uint32)
    m = O.Model(O.make_params(T, D, F, clusters=C), w, f)
    want_l, want_cs = O.classify(m, xs, K, interleaved=True)
    assert np.array_equal(l1[idx].cpu().numpy(), want_l)
    assert np.array_equal(s1[:, idx].cpu().numpy().view(np.uint32), want_cs.view(np.uint32))
    assert len(np.unique(want_l)) == K      # every class wins somewhere: the argmax is not degenerate
    e.close()


def _sharded_cls_worker(rank, world, port, mode, ret):
    import os

    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        T, D, F, K, n = 1000, 8, 32, 10, 6007
        w, f = ddt.synth_model(T, D, F)
        C = ddt.default_clusters(T // K)
        e = ddt.Engine(0)
        e.load_model_multiclass(ddt.make_params(T, D, F, clusters=C), w, f, K, True, rank, world)
        d = e.synth_tuples_device(0, n, F)
        sc = SR.ShardedClassifier.from_engine(e, mode=mode, chunk_rows=2500)
        labels, scores = sc.classify(d)
        torch.cuda.synchronize()
        m = O.Model(O.make_params(T, D, F, clusters=C), w, f)
        want_l, want_cs = O.classify(m, d.cpu().numpy().view(np.uint32), K, interleaved=True, n_devices=world)
        ret[rank] = bool(np.array_equal(scores.cpu().numpy().view(np.uint32), want_cs.view(np.uint32))
                         and np.array_equal(labels.cpu().numpy(), want_l))
        e.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["allreduce", "chain"])
def test_sharded_classifier_two_ranks_on_one_gpu(mode):
    """config 5 tree-sharded over 2 ranks (gloo, both on GPU 0): per-class partial sums combined, then argmax --
    bit-exact with the oracle's 2-device model (a two-term fp32 add is order independent)."""
    import socket

    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_sharded_cls_worker, args=(r, 2, port, mode, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret.get(0) and ret.get(1), dict(ret)
