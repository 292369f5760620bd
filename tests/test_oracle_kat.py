"""Known-answer tests that pin the CPU oracle (oracle/ddt_oracle.c).

The reference (FPGA RTL) has NO tests, golden vectors or fixtures, so these
hand-computed cases -- one per rule of SURVEY 8(a) -- plus an independent numpy restatement
(tests/refimpl.py) and a scikit-learn cross-check are what the oracle is pinned against.
"""
import numpy as np
import pytest

from oracle import oracle as O
from tests import refimpl as R

F32 = np.float32


def bits(x):
    return int(np.array(x, F32).view(np.uint32))


def one_tree(thr, fidx, mr, leaves, F, **kw):
    return O.pack_model(np.array([thr], F32), np.array([fidx]), np.array([mr]), np.array([leaves], F32), F, **kw)


def score1(m, xrow, **kw):
    x = np.zeros((1, O.tuple_lines(m.params.num_features) * 4), np.uint32)
    x[0, : len(xrow)] = np.asarray(xrow, np.uint32)
    return O.score(m, x, **kw)[0]


# ---- A9: comparator -----------------------------------------------------------------------------
def test_less_equal_greater_depth1():
    # go left iff feature < threshold (DTPU.sv:655-657); equality goes RIGHT
    m = one_tree([0.5], [0], [0], [-1.0, 1.0], F=4)
    assert score1(m, [bits(0.25)]) == -1.0
    assert score1(m, [bits(0.5)]) == 1.0
    assert score1(m, [bits(0.75)]) == 1.0


def test_both_negative_is_inverted_in_reference_mode():
    # raw-bit signed-int compare: for two negative floats the order is inverted
    m0 = one_tree([-0.5], [0], [0], [-1.0, 1.0], F=4, cmp_mode=0)
    m1 = one_tree([-0.5], [0], [0], [-1.0, 1.0], F=4, cmp_mode=1)
    assert score1(m0, [bits(-0.75)]) == 1.0   # IEEE says -0.75 < -0.5 (left); the reference goes right
    assert score1(m1, [bits(-0.75)]) == -1.0  # IEEE mode
    assert score1(m0, [bits(-0.25)]) == -1.0  # and the mirror case
    assert score1(m1, [bits(-0.25)]) == 1.0
    # mixed signs agree in both modes
    for mm in (m0, m1):
        assert score1(mm, [bits(-0.75)] if False else [bits(0.25)]) == 1.0


def test_signed_zero():
    m0 = one_tree([0.0], [0], [0], [-1.0, 1.0], F=4, cmp_mode=0)
    m1 = one_tree([0.0], [0], [0], [-1.0, 1.0], F=4, cmp_mode=1)
    assert score1(m0, [0x80000000]) == -1.0  # -0.0 is INT_MIN: "less" than +0.0 in the reference
    assert score1(m1, [0x80000000]) == 1.0   # IEEE: -0 == +0, not less -> right
    assert score1(m0, [0x00000000]) == 1.0


def test_nan_pattern_orders_as_int_in_reference_mode():
    m = one_tree([1.0e30], [0], [0], [-1.0, 1.0], F=4, missing_bits=0x7FC00001)
    assert score1(m, [0x7FC00000]) == 1.0   # positive NaN pattern > any finite positive as int32
    assert score1(m, [0xFFC00000]) == -1.0  # negative NaN pattern is a very negative int32


# ---- A9: missing --------------------------------------------------------------------------------
def test_missing_follows_flag_not_threshold():
    MISS = 0x7FC00000
    mL = one_tree([0.5], [0], [0], [-1.0, 1.0], F=4, missing_bits=MISS)
    mR = one_tree([0.5], [0], [1], [-1.0, 1.0], F=4, missing_bits=MISS)
    assert score1(mL, [MISS]) == -1.0
    assert score1(mR, [MISS]) == 1.0
    # bit-equality only: another NaN payload is NOT missing (DTPU.sv:653)
    assert score1(mL, [0x7FC00001]) == 1.0


def test_missing_reset_value_zero_makes_zero_missing():
    # CSR205 resets to 0 (EngineCSR.sv:167) => feature 0.0 is "missing"
    m = one_tree([0.5], [0], [1], [-1.0, 1.0], F=4, missing_bits=0)
    assert score1(m, [bits(0.0)]) == 1.0    # 0.0 < 0.5 would go left, but it is missing -> flag -> right
    assert score1(m, [bits(0.25)]) == -1.0


# ---- A7/A10: heap addressing --------------------------------------------------------------------
def test_depth3_every_leaf_reachable():
    # node n tests feature n (7 features), threshold 0.5; leaves carry their own index
    thr = [0.5] * 7
    fidx = list(range(7))
    leaves = [float(k) for k in range(8)]
    m = one_tree(thr, fidx, [0] * 7, leaves, F=8)
    for leaf in range(8):
        b = [(leaf >> 2) & 1, (leaf >> 1) & 1, leaf & 1]
        x = [0.0] * 8
        n = 0
        for lvl in range(3):
            x[n] = 0.75 if b[lvl] else 0.25
            n = 2 * n + 1 + b[lvl]
        assert score1(m, [bits(v) for v in x]) == float(leaf)


def test_feature_index_2047():
    m = one_tree([0.5], [2047], [0], [-1.0, 1.0], F=2048)
    x = np.zeros(2048, np.uint32)
    x[2047] = bits(0.75)
    assert score1(m, x) == 1.0
    x[2047] = bits(0.25)
    assert score1(m, x) == -1.0


def test_wire_packing_little_endian():
    # A2: node n -> weights word n; findex entry n -> u16 n; leaves follow the 2^D-1 internal nodes
    m = one_tree([0.1, 0.2, 0.3], [1, 2, 3], [0, 1, 0], [10.0, 11.0, 12.0, 13.0], F=4)
    w = m.wlines.view(F32)
    assert list(w[:7]) == [F32(0.1), F32(0.2), F32(0.3), 10.0, 11.0, 12.0, 13.0] and w[7] == 0.0
    assert list(m.flines[:3]) == [1, 2 | (1 << 13), 3]
    assert m.params.weights_lines_per_tree == 2 and m.params.findex_lines_per_tree == 1


# ---- A11/A12: summation order -------------------------------------------------------------------
def test_empty_slots_add_zero_and_order_is_pairwise_then_sequential():
    rng = np.random.default_rng(1)
    for T, C in [(3, 1), (8, 1), (9, 1), (20, 2), (100, 1), (100, 8), (257, 4), (1000, 8)]:
        leaves = ((rng.random(T) - 0.5) * 0.2).astype(F32)
        lb = leaves.view(np.uint32)
        want = R.reduce_reference_order(lb, C)
        got_native = np.array(O.lib().orc_reduce_device(lb.ctypes.data, T, C, 0), np.uint32).view(F32)
        got_flopoco = np.array(O.lib().orc_reduce_device(lb.ctypes.data, T, C, 1), np.uint32).view(F32)
        assert got_native == want and got_flopoco == want, (T, C)


def test_order_matters_and_is_the_reference_one():
    # three values whose fp32 sum depends on the order: pairwise(8) then sequential
    big, small = F32(1.0), F32(2.0 ** -24)
    leaves = np.array([big, small, small, small, 0, 0, 0, 0], F32)
    # reference: ((1+e)+(e+e)) = (1 + 2e): (1+e)->1 (tie to even), (e+e)=2e, 1+2e = 1+2^-23
    want = F32(1.0) + F32(2.0 ** -23)
    got = np.array(O.lib().orc_reduce_device(leaves.view(np.uint32).ctypes.data, 8, 1, 1), np.uint32).view(F32)
    assert got == want
    assert got != np.cumsum(leaves, dtype=F32)[-1]  # plain left-to-right gives 1.0


def test_multi_device_chain_sum():
    rng = np.random.default_rng(2)
    T, D, Fe = 37, 3, 8
    m = O.gen_model(T, D, Fe)
    x = O.gen_tuples(0, 50, Fe)
    thr, fidx, mr, leaf = R.unpack_model(m.wlines, m.flines, T, D, O.wlpt(D), O.flpt(D))
    lb = R.traverse_all(thr, fidx, mr, leaf, x, m.params.missing_bits)
    for nd in (1, 2, 3, 8):
        want = R.score_reference_order(lb, m.params.clusters_per_tuple, nd)
        got = O.score(m, x, n_devices=nd)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), nd
        # shard partials + chain add reproduce it too
        run = None
        for (b, e) in R.shard_bounds(T, nd):
            part = O.score_shard(m, x, b, e)
            run = part if run is None else (part + run).astype(F32)
        assert np.array_equal(run.view(np.uint32), got.view(np.uint32))


# ---- whole-path cross-checks ---------------------------------------------------------------------
@pytest.mark.parametrize("T,D,Fe,dist", [(8, 4, 16, 0), (8, 4, 16, 1), (100, 6, 28, 0), (33, 8, 32, 1), (3, 12, 64, 1)])
def test_oracle_equals_independent_numpy_restatement(T, D, Fe, dist):
    for cmp_mode in (0, 1):
        m = O.gen_model(T, D, Fe, dist=dist, cmp_mode=cmp_mode)
        n = 203  # not a multiple of 4 (A14 relaxed: any N is accepted)
        x = O.gen_tuples(7, n, Fe, dist=dist, missing_bits=m.params.missing_bits)
        thr, fidx, mr, leaf = R.unpack_model(m.wlines, m.flines, T, D, O.wlpt(D), O.flpt(D))
        lb = R.traverse_all(thr, fidx, mr, leaf, x, m.params.missing_bits, cmp_mode)
        for r in (0, 1, n - 1):
            assert np.array_equal(O.leaves(m, x[r]), lb[r])
            assert np.array_equal(O.leaves_fast(m, x[r]), lb[r])  # the 8-wide walk the batch scorers use
        want = R.score_reference_order(lb, m.params.clusters_per_tuple)
        for mode in (O.SUM_REF_FLOPOCO, O.SUM_REF_NATIVE):
            got, gold = O.score(m, x, sum_mode=mode, want_gold=True)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32))
        exact = lb.view(F32).astype(np.float64).sum(axis=1)
        assert np.allclose(gold, exact, rtol=0, atol=1e-12)
        f64 = O.score(m, x, sum_mode=O.SUM_F64_SEQ)
        assert np.array_equal(f64, np.cumsum(lb.view(F32).astype(np.float64), axis=1)[:, -1].astype(F32))
        # fp32 reference-order sum is within the north-star tolerance of the fp64 gold
        sabs = np.abs(lb.view(F32).astype(np.float64)).sum(axis=1)
        assert np.all(np.abs(got.astype(np.float64) - gold) <= 1e-6 * np.maximum(np.abs(gold), sabs))


def test_dist1_actually_contains_missing_and_negatives():
    x = O.gen_tuples(0, 2000, 16, dist=1)
    assert 0.03 < np.mean(x[:, :16] == 0x7FC00000) < 0.07
    assert np.any(x[:, :16].view(F32) < 0)


def test_cross_check_against_sklearn():
    sk = pytest.importorskip("sklearn.ensemble")
    rng = np.random.default_rng(3)
    n, Fe, D, T = 1500, 10, 5, 12
    X = rng.random((n, Fe)).astype(F32)  # non-negative: reference and IEEE compare agree
    y = (np.sin(X[:, 0] * 6) + X[:, 1] * X[:, 2] + 0.1 * rng.standard_normal(n)).astype(F32)
    rf = sk.RandomForestRegressor(n_estimators=T, max_depth=D, random_state=0).fit(X, y)
    thr = np.zeros((T, (1 << D) - 1), F32)
    fidx = np.zeros((T, (1 << D) - 1), np.int64)
    leaves = np.zeros((T, 1 << D), F32)
    for i, est in enumerate(rf.estimators_):
        t = est.tree_
        t64, fi, lv = R.pad_to_perfect(t.children_left, t.children_right, t.feature, t.threshold,
                                       t.value[:, 0, 0], D)
        thr[i], fidx[i], leaves[i] = R.sklearn_threshold_to_lt(t64), np.maximum(fi, 0), lv.astype(F32)
    m = O.pack_model(thr, fidx, np.zeros_like(fidx, np.uint8), leaves, Fe)
    Xt = rng.random((400, Fe)).astype(F32)
    xl = O.tuples_from_float(Xt)
    for r in range(0, 400, 7):
        got = O.leaves(m, xl[r]).view(F32)
        want = np.array([est.predict(Xt[r:r + 1])[0] for est in rf.estimators_]).astype(F32)
        assert np.array_equal(got, want)
    # and the forest mean equals sum/T within fp32 rounding
    sc = O.score(m, xl, sum_mode=O.SUM_F64_SEQ)
    assert np.allclose(sc / T, rf.predict(Xt), rtol=2e-6, atol=1e-6)


# ---- synthetic generator golden values (computed independently with Python integers) --------------
def _sm64(x):
    M = (1 << 64) - 1
    z = (x + 0x9E3779B97F4A7C15) & M
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & M
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & M
    return z ^ (z >> 31)


def test_generators_match_python_integer_model():
    SX, SM = 0x0DD7000000000001, 0x0DD7000000000002
    for k in (0, 1, 12345, (1 << 40) + 7):
        assert O.lib().orc_splitmix64(k) == _sm64(k)
    Fe, D, T = 28, 6, 5
    x = O.gen_tuples(1000, 3, Fe)
    for r in range(3):
        for j in (0, 13, 27):
            u = (_sm64(SX + (1000 + r) * Fe + j) >> 40) / 16777216.0
            assert x[r, j] == bits(u)
        assert not x[r, 28:].any()
    m = O.gen_model(T, D, Fe)
    thr, fidx, mr, leaf = R.unpack_model(m.wlines, m.flines, T, D, O.wlpt(D), O.flpt(D))
    for i, n in [(0, 0), (4, 62), (2, 17)]:
        g = i * (1 << (D + 1)) + n
        assert fidx[i, n] == _sm64(SM + 3 * g) % Fe
        assert thr[i, n] == bits((_sm64(SM + 3 * g + 1) >> 40) / 16777216.0)
        assert mr[i, n] == _sm64(SM + 3 * g + 2) & 1
    for i, n in [(0, 63), (3, 126)]:
        g = i * (1 << (D + 1)) + n
        u = F32((_sm64(SM + 3 * g + 1) >> 40) / 16777216.0)
        assert leaf[i, n - 63] == bits(F32(F32(u - F32(0.5)) * F32(0.2)))


def test_cpu_baseline_scorer_equals_the_plain_oracle():
    """orc_score_fast (bench.py's cpu_baseline: cache-blocked, branch-free, 8-byte nodes) == orc_score bit for bit."""
    for (T, D, F, rows, dist) in [(1000, 8, 32, 3000, 0), (37, 6, 28, 2049, 1), (9, 3, 5, 130, 1), (100, 6, 28, 63, 1)]:
        m = O.gen_model(T, D, F, dist)
        x = O.gen_tuples(11, rows, F, dist)
        for mode in (O.SUM_REF_NATIVE, O.SUM_F64_SEQ):
            a, b = O.score(m, x, sum_mode=mode), O.score_fast(m, x, sum_mode=mode)
            assert np.array_equal(a.view(np.uint32), b.view(np.uint32)), (T, D, F, mode)
    m = O.gen_model(20, 5, 12, 1, cmp_mode=1)  # modes it does not specialise fall through to the plain scorer
    x = O.gen_tuples(0, 500, 12, 1)
    assert np.array_equal(O.score(m, x, sum_mode=O.SUM_REF_NATIVE).view(np.uint32), O.score_fast(m, x).view(np.uint32))
