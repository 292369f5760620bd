"""Small batches on the plain depth-8 cluster-major kernel (`q16_d8_c8_u4_gl_s2_cm_x`): the launch is cut into SLICES of the image
(csrc/ddt_kernels.hip score_q16_kernel SPLIT, csrc/ddt_engine.cpp cluster_split_of) -- one block per (tile, slice): a slice is a cluster,
whose accumulator the block leaves, or -- for batches too small to fill the chip with (tile, cluster) blocks -- a run of PU groups, whose sums it
leaves one by one; `launch_cm_combine` runs the adds in the reference's order (FPAggregator.v:79-131: per cluster acc <- x + acc; Core.sv:486-541:
the clusters added in order afterwards).  Against the oracle and against the uncut launch, bit for bit, in both
adders, with missing values, ragged row counts, every cluster count, tree counts whose last PU group is partly filled and whose clusters are
unequal; the automatic rule's limit; and the point of it: a 1000-tree call on a batch of a few tiles takes a fraction of the uncut call's time."""
import time

import numpy as np
import pytest

from oracle import oracle as O
import ddt

pytestmark = pytest.mark.gpu
NAME = "q16_d8_c8_u4_gl_s2_cm_x"


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _tuples(n, F, seed, holes):
    x = O.gen_tuples(seed, n, F, dist=0)
    rng = np.random.default_rng(seed)
    for r in rng.integers(0, n, holes):
        x[r, rng.integers(0, F)] = 0x7FC00000  # the default missing pattern: that tile takes the slow image
    return x


@pytest.mark.parametrize("T,clusters", [(1000, 8), (125, 8), (300, 2), (230, 4), (229, 8), (20, 8), (17, 2), (224, 1)])
def test_cut_launch_is_the_uncut_launch_bit_for_bit(T, clusters):
    import torch

    D, F = 8, 32
    m = O.gen_model(T, D, F, dist=1, clusters=clusters)
    e = ddt.Engine(0)
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        e.set_option("variant", ddt.variant_names().index(NAME))
        e.load_model(ddt.make_params(T, D, F, clusters=clusters, sum_mode=sum_mode), m.wlines, m.flines)
        assert e.info().variant_name.decode() == NAME
        for n, holes in ((1, 0), (63, 1), (1024, 0), (5000, 3), (70_001, 9)):
            x = _tuples(n, F, 7 + n % 100, holes)
            d = torch.from_numpy(x.view(np.int32)).cuda()
            want = O.score_fast(m, x, sum_mode=ref)
            got = {}
            for split, groups in ((0, -1), (1, 0), (1, -1), (-1, 3), (-1, 1000)):   # groups: 0 = the slices are the clusters, -1 = automatic, k = k slices
                e.set_option("q16_cluster_split", split)
                e.set_option("q16_split_groups", groups)
                before = e.stats().kernel_launches
                outs = [e.score_device(d) for _ in range(2)]               # back to back: the partial sums' workspace is reused in stream order
                torch.cuda.synchronize()
                cut = split != 0 and (T + 7) // 8 > 1 and (clusters > 1 or groups != 0)
                assert e.stats().kernel_launches - before == (4 if cut else 2), (split, groups, n)
                for o in outs:
                    bad = np.flatnonzero(_bits(o.cpu().numpy()) != _bits(want))
                    assert bad.size == 0, (T, clusters, sum_mode, n, split, groups, bad[:8], bad.size)
                got[(split, groups)] = outs[0].cpu().numpy()
            assert all(np.array_equal(_bits(got[(0, -1)]), _bits(g)) for g in got.values())
    e.close()


@pytest.mark.parametrize("T,D,F,clusters,name", [(1000, 6, 28, 8, "q16_d6_c16_u4_s2"), (300, 6, 32, 4, "q16_d6_c16_u4_s2"), (403, 7, 32, 2, "q16_d7_c8_u4_s2"),
                                                  (290, 5, 20, 1, "q16_d5_c32_u4_s2"), (700, 4, 16, 8, "q16_d4_c64_u8"), (1030, 3, 12, 8, "q16_d3_c128_u8"),
                                                  (300, 8, 48, 4, "q16w_d8_c8_u4_gl_s2_cm_x")])
def test_cut_launch_on_every_depth(T, D, F, clusters, name):
    """Depths 3-7 keep their images in STREAM order (group g belongs to cluster g mod C, Core.sv:291-316): a slice is a run of chunks (2 ... 16 PU
    groups each), every group's sum goes out at the group's place in the image, the combine picks every C-th for a cluster's chain.  And the
    wide depth-8 kernel (cluster-major, 33-64 tuple words).  The engine's own kernel for these shapes, cut automatically."""
    import torch

    m = O.gen_model(T, D, F, dist=1, clusters=clusters)
    e = ddt.Engine(0)
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        e.load_model(ddt.make_params(T, D, F, clusters=clusters, sum_mode=sum_mode), m.wlines, m.flines)
        assert e.info().variant_name.decode() == name
        for n, holes in ((1, 0), (1500, 2), (40_000, 5)):
            x = _tuples(n, F, 11 + n % 100, holes)
            d = torch.from_numpy(x.view(np.int32)).cuda()
            want = O.score_fast(m, x, sum_mode=ref)
            for split, groups in ((0, -1), (-1, -1), (1, 2), (1, 5), (1, 1000)):
                e.set_option("q16_cluster_split", split)
                e.set_option("q16_split_groups", groups)
                before = e.stats().kernel_launches
                got = e.score_device(d)
                torch.cuda.synchronize()
                assert e.stats().kernel_launches - before == (1 if split == 0 else 2), (split, groups, n)
                bad = np.flatnonzero(_bits(got.cpu().numpy()) != _bits(want))
                assert bad.size == 0, (name, T, clusters, sum_mode, n, split, groups, bad[:8], bad.size)
    e.close()


def test_automatic_rule_and_host_buffers():
    import torch

    T, D, F, clusters = 256, 8, 32, 8
    m = O.gen_model(T, D, F, dist=1, clusters=clusters)
    e = ddt.Engine(0)
    e.load_model(ddt.make_params(T, D, F, clusters=clusters), m.wlines, m.flines)
    assert e.info().variant_name.decode() == NAME                          # the engine's own choice for 256 trees
    e.set_option("q16_split_max_tiles", 3)
    for n, cut in ((3072, True), (3073, False)):                           # (automatic: runs of PU groups below a chip's worth of (tile, cluster) blocks)
        x = _tuples(n, F, 3, 2)
        d = torch.from_numpy(x.view(np.int32)).cuda()
        before = e.stats().kernel_launches
        got = e.score_device(d)
        torch.cuda.synchronize()
        assert e.stats().kernel_launches - before == (2 if cut else 1)
        assert np.array_equal(_bits(got.cpu().numpy()), _bits(O.score_fast(m, x)))
    e.set_option("q16_split_max_tiles", 384)
    x = _tuples(40_000, F, 5, 4)                                           # pageable host memory through the feeder's slots
    assert np.array_equal(_bits(e.score(x)), _bits(O.score_fast(m, x)))
    e.close()


def test_a_small_batch_is_faster_cut():
    """1000 trees, one tile: the uncut launch is ONE block walking 1000 trees (~0.35 ms); cut it is a block per PU group on 125 CUs."""
    import torch

    T, D, F = 1000, 8, 32
    w, f = ddt.synth_model(T, D, F, 0)
    e = ddt.Engine(0)
    e.load_model(ddt.make_params(T, D, F), w, f)
    assert e.info().variant_name.decode() == NAME
    d = e.synth_tuples_device(0, 1024, F, 0)
    out = torch.empty(1024, dtype=torch.float32, device=d.device)
    med, res = {}, {}
    for split in (0, -1):
        e.set_option("q16_cluster_split", split)
        for _ in range(5):
            e.score_device(d, out=out)
        torch.cuda.synchronize()
        ts = []
        for _ in range(30):
            t0 = time.perf_counter()
            e.score_device(d, out=out)
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        med[split] = sorted(ts)[len(ts) // 2]
        res[split] = out.cpu().numpy().copy()
    assert np.array_equal(_bits(res[0]), _bits(res[-1]))
    assert med[-1] < 0.35 * med[0], med
    e.close()


@pytest.mark.parametrize("T,D,F,clusters,name", [(512, 12, 32, 4, "q16d_d12_k9_c4_u4_cm"), (70, 12, 3, 8, "q16d_d12_k9_c4_u4_cm"), (100, 10, 28, 1, "q16d_d10_k9_c4_u4_cm"),
                                                  (90, 11, 20, 8, "q16d_d11_k8_c8_u4_cm"), (130, 9, 32, 2, "q16d_d9_k8_c8_u4_cm"), (40, 14, 16, 2, "q16d_d14_k9_c4_u4_cm"),
                                                  (64, 12, 48, 4, "q16dw_d12_k9_c4_u4_cm")])
def test_cut_launch_on_the_deep_kernels(T, D, F, clusters, name):
    """The deep kernels (depth 9-15, csrc/ddt_deep.hip SPLIT): slices of PU groups, every group's sum out at group0 + its place in the launch's
    image, one combine behind the last launch -- also for an ensemble scored in PARTS (512 x d12 x 32 is two, 70 x d12 x 3 three: more than
    38848 thresholds per feature), whose parts then hand no state from launch to launch.  Oracle's bits, both adders, the uncut launch beside it."""
    import torch

    m = O.gen_model(T, D, F, dist=0, clusters=clusters)
    e = ddt.Engine(0)
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        e.load_model(ddt.make_params(T, D, F, clusters=clusters, sum_mode=sum_mode), m.wlines, m.flines)
        assert e.info().variant_name.decode() == name
        for n, holes in ((1, 0), (1500, 2), (9000, 4)):
            x = _tuples(n, F, 13 + n % 100, holes)
            d = torch.from_numpy(x.view(np.int32)).cuda()
            want = O.score_fast(m, x, sum_mode=ref)
            launches = {}
            for split, groups in ((0, -1), (-1, -1), (1, 2), (1, 3), (1, 1000)):
                e.set_option("q16_cluster_split", split)
                e.set_option("q16_split_groups", groups)
                before = e.stats().kernel_launches
                got = [e.score_device(d) for _ in range(2)]
                torch.cuda.synchronize()
                launches[(split, groups)] = (e.stats().kernel_launches - before) // 2
                for g in got:
                    bad = np.flatnonzero(_bits(g.cpu().numpy()) != _bits(want))
                    assert bad.size == 0, (name, T, clusters, sum_mode, n, split, groups, bad[:8], bad.size)
            assert all(v == launches[(0, -1)] + 1 for k, v in launches.items() if k != (0, -1)), launches   # the parts' launches + one combine
    e.close()


@pytest.mark.parametrize("T,K,inter", [(1000, 10, True), (48, 3, False), (210, 7, True)])
def test_cut_launch_over_the_classes_of_a_one_vs_all_model(T, K, inter):
    """BASELINE config 5's kernel ("_p": one block per tile walks every class) on batches of a few tiles: the plain kernel's cut form over the same
    image, the combine per class, the argmax -- sums and labels equal the uncut launch's and the oracle's, both adders."""
    import torch

    D, F = 8, 32
    C = ddt.default_clusters(T // K)
    m = O.gen_model(T, D, F, dist=1, clusters=C)
    e = ddt.Engine(0)
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        e.set_option("variant", -1)
        e.load_model_multiclass(ddt.make_params(T, D, F, clusters=C, sum_mode=sum_mode), m.wlines, m.flines, K, inter)
        e.set_option("variant", ddt.variant_names().index("q16_d8_c8_u4_gl_s2_cm_p"))
        assert e.info().variant_name.decode() == "q16_d8_c8_u4_gl_s2_cm_p"
        for n, holes in ((1, 0), (1500, 2), (50_000, 6)):
            x = _tuples(n, F, 17 + n % 100, holes)
            d = torch.from_numpy(x.view(np.int32)).cuda()
            want_l, want_cs = O.classify_fast(m, x, K, inter, sum_mode=ref)
            for split, groups, launches in ((0, -1, 1), (-1, -1, 3), (1, 3, 3), (1, 60000, 3)):
                e.set_option("q16_cluster_split", split)
                e.set_option("q16_split_groups", groups)
                before = e.stats().kernel_launches
                labels, cs = e.classify_device(d)
                torch.cuda.synchronize()
                assert e.stats().kernel_launches - before == launches, (split, groups, n)
                assert np.array_equal(labels.cpu().numpy(), want_l), (T, K, sum_mode, n, split, groups)
                assert np.array_equal(_bits(cs.cpu().numpy()), _bits(want_cs)), (T, K, sum_mode, n, split, groups)
    e.close()
