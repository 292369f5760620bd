"""The persistent rank-quantised kernel (`q16_d8_c8_u4_gl_s2_cm_p`, csrc/ddt_kernels.hip score_q16p_kernel) against the oracle:
blocks that stay resident and walk MANY tiles (the next rank tile prefetched into registers, the chunk ring continued across
tiles when the chunk count is even, tiles handed out through an atomic ticket), tiles with and without missing values side by
side (fast / slow image per tile), and several ensembles in one pass (the classes of a one-vs-all model: sums + argmax written by
the scoring kernel).  Sizes: more tiles than resident blocks (2 x 256 CUs), so every block switches tiles and the tickets are used."""
import numpy as np
import pytest

from oracle import oracle as O
import ddt

pytestmark = pytest.mark.gpu
NAME = "q16_d8_c8_u4_gl_s2_cm_p"


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _vid(name=NAME):
    return ddt.variant_names().index(name)


def _tuples(n, F, seed, holes):
    """dist-0 tuples (no missing value) with the missing pattern planted in a few rows: most tiles take the fast image, some the slow one"""
    x = O.gen_tuples(seed, n, F, dist=0)
    rng = np.random.default_rng(seed)
    for r in rng.integers(0, n, holes):
        x[r, rng.integers(0, F)] = 0x7FC00000  # the default missing pattern
    return x


@pytest.mark.parametrize("T,clusters,n", [(8, 1, 1_200_003),     # one chunk per tile
                                           (16, 1, 1_100_000),    # two chunks: the ring continues across tiles
                                           (24, 2, 1_150_001),    # three chunks (odd: chunk 0 is requested behind the tile-end barrier), two clusters
                                           (125, 1, 1_300_000),   # a shard of an 8-way job: 16 chunks
                                           (250, 2, 700_001)])
def test_persistent_blocks_over_many_tiles(T, clusters, n):
    import torch

    D, F = 8, 32
    m = O.gen_model(T, D, F, dist=1, clusters=clusters)
    x = _tuples(n, F, 11 + T, 40)
    e = ddt.Engine(0)
    d = torch.from_numpy(x.view(np.int32)).cuda()
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        want = O.score_fast(m, x, sum_mode=ref)
        e.set_option("variant", -1)
        e.load_model(ddt.make_params(T, D, F, clusters=clusters, sum_mode=sum_mode), m.wlines, m.flines)
        for name in (NAME, "q16_d8_c8_u4_gl_s2_cm_x"):              # persistent / the plain launch, both with the pinned read order
            e.set_option("variant", _vid(name))
            assert e.info().variant_name.decode() == name
            for _ in range(2):                                  # twice: the tile counter is zeroed per launch
                got = e.score_device(d)
                torch.cuda.synchronize()
                bad = np.flatnonzero(_bits(got.cpu().numpy()) != _bits(want))
                assert bad.size == 0, (name, T, sum_mode, bad[:8], bad.size)
        e.set_option("variant", _vid())
    # a ragged batch smaller than one tile, and one of exactly two tiles
    for k in (777, 2048):
        got = e.score_device(d[:k])
        torch.cuda.synchronize()
        assert np.array_equal(_bits(got.cpu().numpy()), _bits(want[:k]))
    e.close()


@pytest.mark.parametrize("T,K,inter,n", [(1000, 10, True, 600_000), (30, 3, True, 700_001), (48, 3, False, 530_000)])
def test_all_classes_in_one_launch(T, K, inter, n):
    import torch

    D, F = 8, 32
    C = ddt.default_clusters(T // K)
    m = O.gen_model(T, D, F, dist=1, clusters=C)
    x = _tuples(n, F, 5 + K, 25)
    want_l, want_cs = O.classify(m, x, K, interleaved=inter)
    e = ddt.Engine(0)
    e.load_model_multiclass(ddt.make_params(T, D, F, clusters=C), m.wlines, m.flines, K, inter)
    e.set_option("variant", _vid())                            # (small models would not take a rank-quantised kernel by themselves)
    assert e.info().variant_name.decode() == NAME
    launches = e.stats().kernel_launches
    d = torch.from_numpy(x.view(np.int32)).cuda()
    dl, dcs = e.classify_device(d)
    torch.cuda.synchronize()
    assert e.stats().kernel_launches == launches + 1           # ONE scoring launch for all K classes (and no argmax pass)
    assert np.array_equal(dl.cpu().numpy(), want_l)
    assert np.array_equal(_bits(dcs.cpu().numpy()), _bits(want_cs))
    _, only_cs = e.classify_device(d, want_labels=False)        # the sharded job's call: sums only
    torch.cuda.synchronize()
    assert np.array_equal(_bits(only_cs.cpu().numpy()), _bits(want_cs))
    hl, hs = e.classify(x[:5000], want_scores=True)             # host feeder path
    assert np.array_equal(hl, want_l[:5000]) and np.array_equal(_bits(hs), _bits(want_cs[:, :5000]))
    e.close()


def test_classes_of_unequal_size_fall_back_to_one_launch_per_class():
    import torch

    T, K, D, F, n = 37, 5, 8, 32, 3000                          # interleaved: 8, 8, 7, 7, 7 trees
    m = O.gen_model(T, D, F, dist=1, clusters=1)
    x = O.gen_tuples(3, n, F, dist=1)
    want_l, want_cs = O.classify(m, x, K, interleaved=True)
    e = ddt.Engine(0)
    e.load_model_multiclass(ddt.make_params(T, D, F, clusters=1), m.wlines, m.flines, K, True)
    e.set_option("variant", _vid())
    assert e.info().variant_name.decode() == NAME
    launches = e.stats().kernel_launches
    dl, dcs = e.classify_device(torch.from_numpy(x.view(np.int32)).cuda())
    torch.cuda.synchronize()
    assert e.stats().kernel_launches == launches + K
    assert np.array_equal(dl.cpu().numpy(), want_l) and np.array_equal(_bits(dcs.cpu().numpy()), _bits(want_cs))
    e.close()


@pytest.mark.parametrize("per_class,C,K", [(100, 2, 3), (132, 4, 3), (100, 8, 2), (12, 2, 4), (260, 4, 2)])
def test_partial_pu_group_inside_a_cluster_major_image(per_class, C, K):
    """with more than one cluster the partly filled PU group of a class (trees per class mod 8 in 1..4) is the last group of ITS
    cluster's run in the cluster-major image, not of the image; the one-launch kernel must skip the padding half of THAT group and walk every
    real tree (it used to drop four real trees and walk the padding whenever G > C and G % C != 0)."""
    import torch

    D, F, T, n = 8, 32, per_class * K, 300_000
    m = O.gen_model(T, D, F, dist=1, clusters=C)
    x = _tuples(n, F, 40 + per_class, 15)
    want_l, want_cs = O.classify_fast(m, x, K, True)
    e = ddt.Engine(0)
    e.load_model_multiclass(ddt.make_params(T, D, F, clusters=C), m.wlines, m.flines, K, True)
    e.set_option("variant", _vid())
    assert e.info().variant_name.decode() == NAME
    launches = e.stats().kernel_launches
    dl, dcs = e.classify_device(torch.from_numpy(x.view(np.int32)).cuda())
    torch.cuda.synchronize()
    assert e.stats().kernel_launches == launches + 1
    assert np.array_equal(_bits(dcs.cpu().numpy()), _bits(want_cs)) and np.array_equal(dl.cpu().numpy(), want_l)
    e.close()


def test_unequal_classes_one_launch_per_class_share_no_tile_counter():
    """unequal classes fall back to one "_p" launch per class; those launches take their tiles from ONE ticket counter, so they
    must not overlap on two streams.  More tiles than 2 x resident blocks (the tickets are used), more than two classes (the two-stream
    alternation would apply), every row compared."""
    import torch

    T, K, D, F, n = 37, 3, 8, 32, 2_300_001                      # interleaved: 13, 12, 12 trees
    m = O.gen_model(T, D, F, dist=1, clusters=1)
    x = _tuples(n, F, 77, 50)
    want_l, want_cs = O.classify_fast(m, x, K, True)
    e = ddt.Engine(0)
    e.load_model_multiclass(ddt.make_params(T, D, F, clusters=1), m.wlines, m.flines, K, True)
    e.set_option("variant", _vid())
    assert e.info().variant_name.decode() == NAME
    d = torch.from_numpy(x.view(np.int32)).cuda()
    for _ in range(2):
        launches = e.stats().kernel_launches
        dl, dcs = e.classify_device(d)
        torch.cuda.synchronize()
        assert e.stats().kernel_launches == launches + K
        bad = np.flatnonzero((_bits(dcs.cpu().numpy()) != _bits(want_cs)).any(axis=0))
        assert bad.size == 0, (bad[:8], bad.size)
        assert np.array_equal(dl.cpu().numpy(), want_l)
    e.close()
