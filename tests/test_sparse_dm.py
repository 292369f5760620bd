"""Sparse forests with DENSE MID LEVELS (`sparse_dm<M>_*`, csrc/ddt_sparse.hip, round 5) on the GPU: the M levels right below the top image are
8-byte heap records in global memory (children by index), the dense block of 16-byte records sits M levels lower.  Every forced M and the
engine's own choice against the sparse oracle, bit for bit; tiles with and without missing values; ragged sizes; both adders; and the
A/B switch of the finished walkers' gathers (`sparse_idle_oob`).  The per-node work is the reference's (DTPU.sv:579-720), the sums in the
reference's order (FPAddersReduceTree.sv:94-141, FPAggregator.v:79-131, Core.sv:486-541)."""
import numpy as np
import pytest

from oracle import oracle as O
import ddt

pytestmark = pytest.mark.gpu


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("T,depth,F,full,pm,dist", [(64, 16, 64, 10, 700, 0), (40, 14, 64, 9, 500, 1), (24, 12, 40, 4, 800, 1), (9, 9, 64, 3, 600, 1)])
def test_dense_mid_levels_equal_the_oracle(T, depth, F, full, pm, dist):
    import torch

    sp = O.gen_sparse_model(T, depth, F, full, pm, dist)
    n = 200_003
    x = O.gen_tuples(5, n, F, dist=dist)
    if dist == 0:
        x[::9973, 3] = 0x7FC00000                      # a few tiles with a missing value
    d = torch.from_numpy(x.view(np.int32)).cuda()
    lines, first = np.ascontiguousarray(sp.node_lines, np.uint32), np.ascontiguousarray(sp.first, np.uint64)
    e = ddt.Engine(0)
    e.set_option("sparse_q16", 0)                      # (forests this small would fit u16 ranks: the fp32 family is what has the mid levels)
    e.set_option("sparse_dp", 0)                       # (the dense pair records, tests/test_sparse_dp.py, would take the fuller forests)
    seen = set()
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        want = O.score_sparse_fast(sp, x, sum_mode=ref) if sum_mode == 0 else O.score_sparse(sp, x, sum_mode=ref)
        for dm in (-1, 0, 1):
            e.set_option("sparse_dm", dm)
            e.load_model_sparse(ddt.make_sparse_params(T, depth, F, sum_mode=sum_mode), lines, first)
            name = e.info().variant_name.decode()
            seen.add(name)
            if F == 64:   # (64 features: the choice is K = 8 in two 256-tuple blocks, which has the mid-level siblings)
                assert name.startswith("sparse_dm%d_k8_u8_t256" % dm) if dm > 0 else (dm != 0 or name == "sparse_dk_k8_u8_t256"), (dm, name)
            for oob in (1, 0):
                e.set_option("sparse_idle_oob", oob)
                got = e.score_device(d)
                torch.cuda.synchronize()
                bad = np.flatnonzero(_bits(got.cpu().numpy()) != _bits(want))
                assert bad.size == 0, (name, sum_mode, oob, bad[:8], bad.size)
            for k in (1, 255, 257, 5000):
                got = e.score_device(d[:k])
                torch.cuda.synchronize()
                assert np.array_equal(_bits(got.cpu().numpy()), _bits(want[:k])), (name, k)
    assert len(seen) >= (2 if F == 64 else 1)
    e.close()
