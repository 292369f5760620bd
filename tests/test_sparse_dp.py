"""Sparse forests with DENSE PAIR RECORDS (`sparse_dp_*`, csrc/ddt_sparse.hip, round 5) on the GPU: the two levels right below the top image are
one block of 16-byte records {key of the level-K node, keys of its two children, the three feature numbers as bytes + their missing directions}
-- ONE gather decides two levels -- and the dense block of ordinary records sits at level K + 2.  Forced and the engine's own choice against the
sparse oracle, bit for bit; tiles with and without missing values (on the features the pair records test); ragged sizes; both adders; the A/B
switches of the finished walkers' gathers and of the loop's last round.  The per-node work is the reference's (DTPU.sv:579-720), the sums in the
reference's order (FPAddersReduceTree.sv:94-141, FPAggregator.v:79-131, Core.sv:486-541)."""
import numpy as np
import pytest

from oracle import oracle as O
import ddt

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("T,depth,F,full,pm,dist", [(64, 16, 64, 10, 700, 0), (40, 14, 64, 11, 500, 1), (24, 12, 40, 4, 800, 1), (9, 9, 64, 3, 600, 1), (17, 10, 100, 10, 0, 1)])
def test_dense_pair_records_equal_the_oracle(T, depth, F, full, pm, dist):
    import torch

    sp = O.gen_sparse_model(T, depth, F, full, pm, dist)
    n = 200_003
    x = O.gen_tuples(5, n, F, dist=dist)
    if dist == 0:
        x[::9973, 3] = 0x7FC00000                      # a few tiles with a missing value
    else:
        x[::331, :] = np.where(np.arange(F)[None, :] % 3 == 0, np.uint32(0x7FC00000), x[::331, :])   # missing values on a third of the features: every level sees them
    d = torch.from_numpy(x.view(np.int32)).cuda()
    lines, first = np.ascontiguousarray(sp.node_lines, np.uint32), np.ascontiguousarray(sp.first, np.uint64)
    e = ddt.Engine(0)
    e.set_option("sparse_q16", 0)                      # (forests this small would fit u16 ranks: the fp32 family is what has the pair records)
    seen = set()
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        want = O.score_sparse_fast(sp, x, sum_mode=ref) if sum_mode == 0 else O.score_sparse(sp, x, sum_mode=ref)
        for dp in (1, -1):
            e.set_option("sparse_dp", dp)
            e.load_model_sparse(ddt.make_sparse_params(T, depth, F, sum_mode=sum_mode), lines, first)
            name = e.info().variant_name.decode()
            seen.add(name)
            assert dp < 0 or F > 72 or F <= 16 or name.startswith("sparse_dp_k"), (dp, name)     # forced: the two-block geometries with K = 7 .. 9 (17 .. 72 features) have the sibling
            for oob, peel in ((1, 1), (1, 0), (0, 1)):
                e.set_option("sparse_idle_oob", oob)
                e.set_option("sparse_peel_last", peel)
                got = e.score_device(d)
                torch.cuda.synchronize()
                bad = np.flatnonzero(_bits(got.cpu().numpy()) != _bits(want))
                assert bad.size == 0, (name, sum_mode, oob, peel, bad[:8], bad.size)
            e.set_option("sparse_idle_oob", 1)
            e.set_option("sparse_peel_last", 1)
            for k in (1, 255, 257, 5000):
                got = e.score_device(d[:k])
                torch.cuda.synchronize()
                assert np.array_equal(_bits(got.cpu().numpy()), _bits(want[:k])), (name, k)
    assert F > 72 or any(s.startswith("sparse_dp_k") for s in seen)
    e.close()


def test_dense_pair_records_with_classes_and_tree_shards():
    """one-vs-all classes and tree shards (an engine loads a subset of the stream's trees) over pair-record images: every class / shard packs
    its own blocks; per-class sums bit-exact, labels exact; the chain add of two class shards"""
    import torch

    e = ddt.Engine(0)
    e.set_option("sparse_q16", 0)
    e.set_option("sparse_dp", 1)
    for (T, D, F, K, inter) in [(60, 13, 64, 3, True), (48, 12, 64, 4, False)]:
        s = O.gen_sparse_model(T, D, F, 10, 650, 1, clusters=1)
        x = O.gen_tuples(5, 3001, F, 1)
        x[::17, 2] = s.params.missing_bits
        want_l, want_s = O.classify_sparse(s, x, K, inter)
        p = ddt.make_sparse_params(T, D, F, clusters=1)
        e.load_model_sparse(p, s.node_lines, s.first, 0, 1, K, inter)
        assert e.info().variant_name.decode().startswith("sparse_dp_k")
        d = torch.from_numpy(x.view(np.int32)).cuda()
        gl, gs = e.classify_device(d)
        torch.cuda.synchronize()
        assert np.array_equal(gl.cpu().numpy(), want_l) and np.array_equal(_bits(gs.cpu().numpy()), _bits(want_s)), (T, K, inter)
        parts = []
        for g in range(2):
            e.load_model_sparse(p, s.node_lines, s.first, g, 2, K, inter)
            assert e.info().variant_name.decode().startswith("sparse_dp_k")
            parts.append(e.classify_device(d, want_labels=False)[1])
        comb = torch.stack([e.chain_sum_device(torch.stack([parts[0][k], parts[1][k]])) for k in range(K)])
        lab = e.argmax_device(comb.contiguous())
        wl2, ws2 = O.classify_sparse(s, x, K, inter, n_devices=2)
        assert np.array_equal(lab.cpu().numpy(), wl2) and np.array_equal(_bits(comb.cpu().numpy()), _bits(ws2))
    e.close()


@pytest.mark.parametrize("T,depth,F,full,pm", [(64, 16, 64, 10, 700), (40, 13, 72, 9, 500), (30, 12, 64, 11, 300), (24, 12, 20, 11, 300)])
def test_dense_pair_records_rank_quantised(T, depth, F, full, pm):
    """the rank-quantised family (`sparse_qp_*`: thresholds -> ranks, features -> the u16 tiles of the q16 pre-pass): the same pair records with
    ranks as keys; missing values (rank 0xFFFF); both adders; ragged sizes"""
    import torch

    sp = O.gen_sparse_model(T, depth, F, full, pm, 1)
    n = 150_001
    x = O.gen_tuples(7, n, F, dist=1)
    x[::29, :] = np.where(np.arange(F)[None, :] % 4 == 1, np.uint32(0x7FC00000), x[::29, :])
    d = torch.from_numpy(x.view(np.int32)).cuda()
    lines, first = np.ascontiguousarray(sp.node_lines, np.uint32), np.ascontiguousarray(sp.first, np.uint64)
    e = ddt.Engine(0)
    e.set_option("sparse_dp", 1)
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        want = O.score_sparse_fast(sp, x, sum_mode=ref) if sum_mode == 0 else O.score_sparse(sp, x, sum_mode=ref)
        e.load_model_sparse(ddt.make_sparse_params(T, depth, F, sum_mode=sum_mode), lines, first)
        name = e.info().variant_name.decode()
        assert name.startswith("sparse_qp_k") or F <= 32, name      # (K = 10, which up to 32 features reach, has no pair-record form: measured slower)
        for k in (n, 1, 1023, 1025, 5000):
            got = e.score_device(d[:k])
            torch.cuda.synchronize()
            bad = np.flatnonzero(_bits(got.cpu().numpy()) != _bits(want[:k]))
            assert bad.size == 0, (name, sum_mode, k, bad[:8], bad.size)
    e.close()
