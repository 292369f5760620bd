"""Sparse forests on 32-BIT RANKS with pair records on every deep level (`sparse_r_*`, csrc/ddt_sparse_r.hip, round 6) on the GPU, through the
C-ABI against the sparse oracle, bit for bit: a node is one word {rank : 17 | feature : 7 | flags : 8}, the feature tile holds rank(x) << 15 | 0x7FFF
written by the rank32 pre-pass (directory out of LDS + one 16-byte gather of the key block), a 16-byte record {node, left child, right child,
pointer} decides two levels per gather.  The engine's own choice (asserted by name) and the forced one; tiles with and without missing values;
ragged sizes; both comparators; all three sums; tables below and above one key block per directory entry; classes and tree shards; the host
feeder.  The per-node work is the reference's (DTPU.sv:579-720), the sums in the reference's order (FPAddersReduceTree.sv:94-141,
FPAggregator.v:79-131, Core.sv:486-541)."""
import numpy as np
import pytest

from oracle import oracle as O
import ddt

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("T,depth,F,full,pm,dist,auto", [(136, 16, 64, 10, 700, 0, True), (70, 14, 20, 6, 700, 1, True), (40, 14, 64, 11, 500, 1, False),
                                                          (24, 12, 40, 4, 800, 1, False), (9, 9, 64, 3, 600, 1, False), (17, 10, 100, 10, 0, 1, False),
                                                          (12, 20, 100, 2, 850, 1, False), (3, 2, 5, 1, 500, 1, False)])
def test_pair_records_on_32_bit_ranks_equal_the_oracle(T, depth, F, full, pm, dist, auto):
    import torch

    sp = O.gen_sparse_model(T, depth, F, full, pm, dist)
    n = 120_003
    x = O.gen_tuples(5, n, F, dist=dist)
    if dist == 0:
        x[::9973, 3] = 0x7FC00000                      # a few tiles with a missing value
    else:
        x[::331, :] = np.where(np.arange(x.shape[1])[None, :] % 3 == 0, np.uint32(0x7FC00000), x[::331, :])   # missing values on a third of the features: every level sees them
    # values ON thresholds and right beside them (the rank must count keys <= x: an off-by-one flips exactly these rows)
    thr = sp.node_lines[:, 0][: 4000]
    fj = sp.node_lines[:, 1][: 4000] & 0x7FF
    rows = np.arange(thr.size) * 7 + 11
    x[rows, fj] = thr
    x[rows + 1, fj] = thr - 1
    x[rows + 2, fj] = thr + 1
    d = torch.from_numpy(x.view(np.int32)).cuda()
    lines, first = np.ascontiguousarray(sp.node_lines, np.uint32), np.ascontiguousarray(sp.first, np.uint64)
    e = ddt.Engine(0)
    seen = set()
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO), (1, O.SUM_F64_SEQ)):
        want = O.score_sparse_fast(sp, x, sum_mode=ref) if sum_mode == 0 else O.score_sparse(sp, x[:30_000], sum_mode=ref)
        for r32 in (-1, 1):
            e.set_option("sparse_r32", r32)
            e.load_model_sparse(ddt.make_sparse_params(T, depth, F, sum_mode=sum_mode), lines, first)
            name = e.info().variant_name.decode()
            seen.add(name)
            if r32 > 0 or auto:
                assert name.startswith("sparse_r_k"), (r32, name)
                assert not e.info().fallback_kernel
            else:
                assert not name.startswith("sparse_r_k"), name   # shallow / few trees per tuple word: the pre-pass does not pay (ddt_sparse_host.cpp)
            for k in ((n, 1, 255, 257, 5000) if sum_mode == 0 else (30_000,)):
                got = e.score_device(d[:k])
                torch.cuda.synchronize()
                bad = np.flatnonzero(_bits(got.cpu().numpy()) != _bits(want[:k]))
                assert bad.size == 0, (name, sum_mode, k, bad[:8], bad.size)
        if sum_mode == 0:
            assert np.array_equal(_bits(e.score(x[:70_001])), _bits(want[:70_001]))   # the host feeder: workspace slots of its streams
    assert any(s.startswith("sparse_r_k") for s in seen)
    e.close()


def test_ieee_comparator_and_key_blocks_of_eight(monkeypatch):
    """cmp_mode 1 (IEEE `<` through the order-preserving key: negative values, -0, NaN features), a feature with ~100 k distinct thresholds (a
    directory of 25 k entries in the pre-pass), and blocks of EIGHT keys (two gathers per value: what tables beyond 4 x 32767 keys get; a 17-bit rank
    ends at 131,070, so the block size is forced here -- `DDT_R32_BLK_LOG2`, read when the model is loaded)"""
    import torch

    e = ddt.Engine(0)
    e.set_option("sparse_r32", 1)
    for (T, depth, F, full, pm, cmp_mode, blk) in [(40, 15, 12, 5, 750, 1, 0), (2, 16, 1, 14, 980, 0, 0), (4, 16, 2, 14, 980, 1, 3)]:
        if blk:
            monkeypatch.setenv("DDT_R32_BLK_LOG2", str(blk))
        else:
            monkeypatch.delenv("DDT_R32_BLK_LOG2", raising=False)
        sp = O.gen_sparse_model(T, depth, F, full, pm, 1, cmp_mode=cmp_mode)
        n = 50_001
        x = O.gen_tuples(9, n, F, dist=1)
        xf = x.view(np.float32)
        xf[::13, 0] = -xf[::13, 0]                   # negative features
        x[5::1001, 0] = 0x80000000                   # -0
        x[7::1003, 0] = 0x7FC00001                   # a NaN that is not the missing pattern
        x[::577, 0] = sp.params.missing_bits
        d = torch.from_numpy(x.view(np.int32)).cuda()
        want = O.score_sparse(sp, x)
        e.load_model_sparse(ddt.make_sparse_params(T, depth, F, cmp_mode=cmp_mode, sum_mode=2), sp.node_lines, sp.first)
        assert e.info().variant_name.decode().startswith("sparse_r_k")
        got = e.score_device(d)
        torch.cuda.synchronize()
        bad = np.flatnonzero(_bits(got.cpu().numpy()) != _bits(want))
        assert bad.size == 0, (T, depth, F, cmp_mode, bad[:8], bad.size)
    monkeypatch.delenv("DDT_R32_BLK_LOG2", raising=False)
    e.close()


def test_classes_and_tree_shards():
    """one-vs-all classes (the batch is ranked once, by the first class's launch) and tree shards over pair-record images; per-class sums
    bit-exact, labels exact; the chain add of two class shards (ResultsCombiner.sv:292-311)"""
    import torch

    e = ddt.Engine(0)
    e.set_option("sparse_r32", 1)
    for (T, D, F, K, inter) in [(60, 13, 64, 3, True), (48, 15, 30, 4, False)]:
        s = O.gen_sparse_model(T, D, F, 8, 650, 1, clusters=1)
        x = O.gen_tuples(5, 3001, F, 1)
        x[::17, 2] = s.params.missing_bits
        want_l, want_s = O.classify_sparse(s, x, K, inter)
        p = ddt.make_sparse_params(T, D, F, clusters=1)
        e.load_model_sparse(p, s.node_lines, s.first, 0, 1, K, inter)
        assert e.info().variant_name.decode().startswith("sparse_r_k")
        d = torch.from_numpy(x.view(np.int32)).cuda()
        gl, gs = e.classify_device(d)
        torch.cuda.synchronize()
        assert np.array_equal(gl.cpu().numpy(), want_l) and np.array_equal(_bits(gs.cpu().numpy()), _bits(want_s)), (T, K, inter)
        parts = []
        for g in range(2):
            e.load_model_sparse(p, s.node_lines, s.first, g, 2, K, inter)
            assert e.info().variant_name.decode().startswith("sparse_r_k")
            parts.append(e.classify_device(d, want_labels=False)[1])
        comb = torch.stack([e.chain_sum_device(torch.stack([parts[0][k], parts[1][k]])) for k in range(K)])
        lab = e.argmax_device(comb.contiguous())
        wl2, ws2 = O.classify_sparse(s, x, K, inter, n_devices=2)
        assert np.array_equal(lab.cpu().numpy(), wl2) and np.array_equal(_bits(comb.cpu().numpy()), _bits(ws2))
    e.close()


@pytest.mark.parametrize("T,D,F,clusters", [(512, 16, 64, 4), (130, 14, 20, 1), (200, 13, 100, 8), (70, 15, 30, 2)])
def test_small_batches_are_cut_into_slices_of_C_groups(T, D, F, clusters):
    """A batch of a few tiles: grid (tiles, slices of C consecutive PU groups); within a slice every cluster's accumulator takes ONE group's sum (in
    stream order group g belongs to cluster g mod C, Core.sv:291-316), the epilogue lets the ring out and `cm_combine_kernel` runs the adds in the
    reference's order.  The uncut launch beside it, the oracle's bits, both adders, missing values, ragged row counts."""
    import torch

    sp = O.gen_sparse_model(T, D, F, 8, 700, 1, clusters=clusters)
    e = ddt.Engine(0)
    e.set_option("sparse_r32", 1)
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        e.load_model_sparse(ddt.make_sparse_params(T, D, F, clusters=clusters, sum_mode=sum_mode), sp.node_lines, sp.first)
        assert e.info().variant_name.decode().startswith("sparse_r_k")
        for n in (1, 700, 20_001):
            x = O.gen_tuples(3 + n % 7, n, F, dist=1)
            x[::97, 1] = sp.params.missing_bits
            d = torch.from_numpy(x.view(np.int32)).cuda()
            want = O.score_sparse_fast(sp, x, sum_mode=ref) if hasattr(O, "score_sparse_fast") else O.score_sparse(sp, x, sum_mode=ref)
            for split, launches in ((0, 1), (-1, 2), (1, 2)):
                e.set_option("q16_cluster_split", split)
                before = e.stats().kernel_launches
                got = [e.score_device(d) for _ in range(2)]
                torch.cuda.synchronize()
                assert e.stats().kernel_launches - before == 2 * launches, (split, n)
                for g in got:
                    bad = np.flatnonzero(_bits(g.cpu().numpy()) != _bits(want))
                    assert bad.size == 0, (T, D, clusters, sum_mode, n, split, bad[:8], bad.size)
    e.close()
