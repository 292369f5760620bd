"""The tree -> PU / cluster SCHEDULE that fixes the reference's fp32 summation order, pinned to the reference's own RTL.

tests/golden/schedule_rtl_vectors.npz was produced by tests/golden/make_schedule_golden.py, which EXECUTES the text of
rtl/DTEngine/Core.sv:167-245,291-372,503-541 and core/RLS.v:36-62 cycle by cycle (a procedural-Verilog interpreter; one
documented repair of the unreachable IDLE -> PROG_MODE edge, see that script).  For every tree it holds the
cluster-enable mask and the PU number its lines were stamped with, for every tuple its cluster mask, and the order in
which cluster partial sums enter the final accumulator.

This test rebuilds the summation from those placements alone -- leaves of the trees that landed in (cluster, PU, slot),
8-way adder tree over the PUs (orc_tree8, itself pinned to FPAddersReduceTree.sv), slot accumulation and cluster
accumulation (orc_aggregate, pinned to FPAggregator.v) in the order the RTL walk produced -- and holds the oracle's
one-line statement of the same thing (orc_reduce_device: tree i -> PU i % 8, group g = i / 8 -> cluster g % C, slot
g / C, clusters added 0..C-1) to it, bit for bit, on leaves whose sum depends on the order."""
import os

import numpy as np
import pytest

from oracle import oracle as O

VEC = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "schedule_rtl_vectors.npz")


@pytest.fixture(scope="module")
def vec():
    return np.load(VEC)


def _cases(v):
    return [tuple(int(x) for x in c) for c in v["cases"]]


def test_weights_and_feature_indexes_of_a_tree_land_in_the_same_place(vec):
    for C, T in _cases(vec):
        w, f = vec[f"w_{C}_{T}"], vec[f"f_{C}_{T}"]
        assert np.array_equal(w, f), (C, T)            # (cluster mask, PU) per tree, both streams
        assert np.array_equal(w[:, 1], np.arange(T) % 8)  # tree i -> PU i % 8 (Core.sv:352-357)


def test_tree_groups_rotate_through_the_clusters_of_every_replica(vec):
    for C, T in _cases(vec):
        w = vec[f"w_{C}_{T}"]
        for i in range(T):
            g = i // 8
            want = 0
            for r in range(8 // C):                   # 8 / C model replicas; group g -> cluster g % C of each
                want |= 1 << (r * C + g % C)
            assert int(w[i, 0]) == want, (C, T, i)


def test_tuples_and_results_walk_the_replicas_round_robin(vec):
    for C, T in _cases(vec):
        t, r = vec[f"t_{C}_{T}"], vec[f"r_{C}_{T}"]
        for k, mask in enumerate(t):
            base = (k * C) % 8
            assert int(mask) == sum(1 << (base + j) for j in range(C)), (C, k)
        # the accumulator takes clusters base+0 .. base+C-1 of tuple k, raising `last` on the C-th (Core.sv:503-541)
        for n, (cl, last) in enumerate(r):
            k, j = divmod(n, C)
            assert int(cl) == ((k * C) % 8 + j) % 8 and int(last) == int(j == C - 1), (C, n)


def _leaves(T, seed):
    rng = np.random.default_rng(seed)
    mant = rng.uniform(0.5, 1.0, T)
    expo = rng.integers(-12, 13, T)                   # wide dynamic range: the rounded sum depends on the order
    sign = rng.choice([-1.0, 1.0], T)
    return (sign * mant * 2.0 ** expo).astype(np.float32)


def test_summation_order_rebuilt_from_the_rtl_placement_equals_the_oracle(vec):
    L = O.lib()
    for C, T in _cases(vec):
        w, t, r = vec[f"w_{C}_{T}"], vec[f"t_{C}_{T}"], vec[f"r_{C}_{T}"]
        # what every (cluster, PU) holds, in arrival order = slot order (DTPU.sv:282-354 writes trees line-sequentially)
        held = {}
        for i in range(T):
            for c in range(8):
                if (int(w[i, 0]) >> c) & 1:
                    held.setdefault((c, int(w[i, 1])), []).append(i)
        for trial in range(4):
            leaves = _leaves(T, 100 * C + trial).view(np.uint32)
            want = L.orc_reduce_device(O._p(np.ascontiguousarray(leaves)), T, C, 1)
            for k in (0, 1, 5):                       # tuples in different replicas see the same order
                order = [int(r[k * C + j, 0]) for j in range(C)]
                assert sorted(order) == [c for c in range(8) if (int(t[k]) >> c) & 1]
                cluster_sums = []
                for c in order:
                    slots = max(len(held.get((c, pu), [])) for pu in range(8))
                    per_slot = []
                    for s in range(slots):
                        l8 = np.zeros(8, np.uint32)
                        for pu in range(8):
                            trees = held.get((c, pu), [])
                            if s < len(trees):
                                l8[pu] = leaves[trees[s]]
                        per_slot.append(L.orc_tree8(O._p(l8)))
                    cluster_sums.append(L.orc_aggregate(O._p(np.array(per_slot, np.uint32)), len(per_slot)))
                got = L.orc_aggregate(O._p(np.array(cluster_sums, np.uint32)), len(cluster_sums))
                assert got == want, (C, T, trial, k)
        # every tree is held exactly once by the clusters of one tuple
        for k in (0, 3):
            seen = sorted(i for c in range(8) if (int(t[k]) >> c) & 1 for pu in range(8) for i in held.get((c, pu), []))
            assert seen == list(range(T)), (C, T, k)
