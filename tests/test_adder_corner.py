"""where the reference's FloPoCo adder is NOT the IEEE-754 add on normal operands, and what each sum_mode does there.

rtl/DTEngine/common/FPAdder_2cycles_latency.v:325-326 -- `shiftedOut = (expDiff >= 25)` forces the alignment shift to 26 already at an
exponent difference of exactly 25.  Consequence: an effective subtraction whose larger operand is an exact power of two, exponents 25
apart, smaller mantissa != 0 returns the larger operand unchanged; IEEE round-to-nearest-even returns the float just below it.
    2^-4 + (-1.5 * 2^-29):  reference / oracle 0x3D800000,  IEEE 0x3D7FFFFF
The oracle (orc_fp34_add) reproduces the RTL (pinned by tests/golden/fpadder_rtl_vectors.npz).  The product:
    sum_mode 0 = reference ORDER with IEEE adds  -> equals the oracle's SUM_REF_NATIVE bit for bit, 1 ulp from the RTL in this case
    sum_mode 2 = reference order with the reference adder (the case reproduced) -> equals SUM_REF_FLOPOCO bit for bit
Both claims are held here on crafted models that put the case at every adder of the path (8-way tree, slot accumulator, cluster sum,
inter-device hop) and on "corner-rich" random models; on the CPU through the real host code + the CPU model of the kernels
(tests/mock_hip, same ref_add_exact() the kernels call), on the GPU through the C-ABI for every kernel variant that takes the model.
"""
import ctypes as C

import numpy as np
import pytest

import ddt
from oracle import oracle as O

BIG, SMALL = np.float32(2.0 ** -4), np.float32(-1.5 * 2.0 ** -29)
RTL_BITS, IEEE_BITS = 0x3D800000, 0x3D7FFFFF


def _const_tree_model(tree_values, D, F, clusters):
    """T perfect trees of depth D whose leaves are all tree_values[i]: every tuple collects exactly these T values."""
    T, nint = len(tree_values), (1 << D) - 1
    thr = np.full((T, nint), 0.5, np.float32)
    fidx = np.zeros((T, nint), np.uint16)
    leaves = np.repeat(np.asarray(tree_values, np.float32)[:, None], 1 << D, axis=1)
    return O.pack_model(thr, fidx, np.zeros((T, nint), np.uint8), leaves, F, clusters=clusters)


def _crafted(where, D=8, F=32):
    """The case at one adder of the path.  -> (model, n_devices)"""
    v = [0.0] * 16
    if where == "tree8":        # l0 + l1 inside one PU group (FPAddersReduceTree.sv:94-141)
        v[0], v[1] = BIG, SMALL
        return _const_tree_model(v, D, F, 1), 1
    if where == "tree8_level2":  # (l0+l1) + (l2+l3)
        v[0], v[2] = BIG, SMALL
        return _const_tree_model(v, D, F, 1), 1
    if where == "slot":         # acc <- s_g + acc on one cluster, two slots (FPAggregator.v:124-131)
        v[0], v[8] = BIG, SMALL
        return _const_tree_model(v, D, F, 1), 1
    if where == "cluster":      # sequential add over the clusters (Core.sv:486-541)
        v[0], v[8] = BIG, SMALL
        return _const_tree_model(v, D, F, 2), 1
    if where == "hop":          # device 1's partial added to device 0's (ResultsCombiner.sv:292-311)
        v[0], v[8] = BIG, SMALL
        return _const_tree_model(v, D, F, 1), 2
    raise KeyError(where)


WHERE = ["tree8", "tree8_level2", "slot", "cluster", "hop"]


def _corner_rich(T, D, F, seed):
    """Leaves from {+-2^e} u {+-(1+u) 2^(e-25)} u {0}: partial sums are powers of two next to addends 25 exponents below."""
    rng = np.random.default_rng(seed)
    nint, nleaf = (1 << D) - 1, 1 << D
    thr = rng.random((T, nint)).astype(np.float32)
    fidx = rng.integers(0, F, (T, nint)).astype(np.uint16)
    kind = rng.random((T, nleaf))
    e = rng.integers(-6, -1, (T, nleaf))
    sign = np.where(rng.random((T, nleaf)) < 0.5, -1.0, 1.0)
    pow2 = sign * np.exp2(e.astype(np.float64))
    tiny = sign * (1.0 + rng.integers(1, 1 << 23, (T, nleaf)) / float(1 << 23)) * np.exp2(e.astype(np.float64) - 25.0)
    leaves = np.where(kind < 0.3, pow2, np.where(kind < 0.7, tiny, 0.0)).astype(np.float32)
    return O.pack_model(thr, fidx, np.zeros((T, nint), np.uint8), leaves, F, clusters=O.default_clusters(T))


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _params(m, sum_mode):
    p = m.params
    return ddt.make_params(p.num_trees, p.num_levels, p.num_features, p.missing_bits, p.cmp_mode, p.clusters_per_tuple, sum_mode)


# ------------------------------------------------------------------------------------------------- oracle: the case exists
def test_the_case_in_the_oracle_adder():
    L = O.lib()
    assert L.orc_fpadd_bits(int(_bits(BIG).item()), int(_bits(SMALL).item())) == RTL_BITS      # the RTL (pinned oracle)
    assert int(_bits(np.float32(BIG + SMALL)).item()) == IEEE_BITS                           # IEEE-754 RNE
    # ... and nothing else differs on normal operands whose result is normal: random pairs + a sweep of exponent gaps
    rng = np.random.default_rng(1)
    a = rng.integers(0x00800000, 0x7F000000, 400_000, dtype=np.uint32) | (rng.integers(0, 2, 400_000, dtype=np.uint32) << 31)
    gap = rng.integers(0, 30, 400_000)
    b = ((a >> 23) & 0xFF).astype(np.int64) - gap
    ok = b > 0
    b = (np.where(ok, b, 1).astype(np.uint32) << 23) | rng.integers(0, 1 << 23, 400_000, dtype=np.uint32) | (rng.integers(0, 2, 400_000, dtype=np.uint32) << 31)
    a[::3] &= 0xFF800000  # exact powers of two as the larger operand
    out = np.zeros_like(a)
    L.orc_fpadd_bits_batch(a.ctypes.data, b.ctypes.data, out.ctypes.data, a.size)
    ieee = _bits(a.view(np.float32) + b.view(np.float32))
    normal_result = ((ieee >> 23) & 0xFF != 0) & ((ieee >> 23) & 0xFF != 0xFF)
    differs = (out != ieee) & normal_result
    ea, eb, ma, mb = (a >> 23) & 0xFF, (b >> 23) & 0xFF, a & 0x7FFFFF, b & 0x7FFFFF
    corner = ((a ^ b) >> 31 == 1) & (((ea.astype(np.int64) - eb == 25) & (ma == 0) & (mb != 0)) | ((eb.astype(np.int64) - ea == 25) & (mb == 0) & (ma != 0)))
    assert differs.sum() > 1000 and np.array_equal(differs, corner & normal_result)
    big = np.where(ea >= eb, a, b)
    assert np.array_equal(out[differs], big[differs])  # the larger operand, unchanged


@pytest.mark.parametrize("where", WHERE)
def test_the_case_in_the_oracle_path(where):
    m, nd = _crafted(where)
    x = O.gen_tuples(0, 8, 32)
    rtl = O.score(m, x, sum_mode=O.SUM_REF_FLOPOCO, n_devices=nd)
    ieee = O.score(m, x, sum_mode=O.SUM_REF_NATIVE, n_devices=nd)
    assert set(_bits(rtl)) == {RTL_BITS} and set(_bits(ieee)) == {IEEE_BITS}


# ------------------------------------------------------------------------------------------------- CPU: host code + kernel model
@pytest.fixture(scope="module")
def mock():
    from tests.test_engine_mock import _build
    return _build("libddt_host_mock.so")


def _mock_score(L, m, x, sum_mode, variant=None, shard=(0, 1)):
    from tests.test_engine_mock import _engine, _load
    L.mock_reset(0, 0, 8)
    e = _engine(L)
    _load(L, e, m, _params(m, sum_mode), variant, shard)
    out = np.full(x.shape[0], np.nan, np.float32)
    assert L.ddt_score(e, x.ctypes.data, x.shape[0], out.ctypes.data) == 0, L.ddt_last_error(e)
    L.ddt_destroy(e)
    return out


@pytest.mark.parametrize("where", WHERE[:-1])
@pytest.mark.parametrize("variant", [None, "q16_d8_c8_u4_gl_s2_cm_x", "q16_d8_c8_u4_gl_s2", "d8_t1024_r1_c4_u4_dma_f"])
def test_host_model_crafted(mock, where, variant):
    m, _ = _crafted(where)
    x = O.gen_tuples(0, 70, 32)
    assert set(_bits(_mock_score(mock, m, x, 0, variant))) == {IEEE_BITS}   # reference order, IEEE adds
    assert set(_bits(_mock_score(mock, m, x, 2, variant))) == {RTL_BITS}    # the reference adder


@pytest.mark.parametrize("T,D,F,seed", [(64, 3, 8, 0), (200, 6, 28, 1), (37, 8, 32, 2)])
def test_host_model_corner_rich(mock, T, D, F, seed):
    m = _corner_rich(T, D, F, seed)
    x = O.gen_tuples(0, 3000, F)
    rtl, ieee = O.score(m, x, sum_mode=O.SUM_REF_FLOPOCO), O.score(m, x, sum_mode=O.SUM_REF_NATIVE)
    assert (_bits(rtl) != _bits(ieee)).mean() > 0.02          # the case really occurs in these sums
    assert np.array_equal(_bits(_mock_score(mock, m, x, 2)), _bits(rtl))
    assert np.array_equal(_bits(_mock_score(mock, m, x, 0)), _bits(ieee))


def test_domain_check_bounds(mock):
    """+0 and normal leaves in [2^-102, 2^96) are the domain on which sum_mode 2 is the reference adder; the loader holds it."""
    from tests.test_engine_mock import _engine
    for value, ok in [(0.0, True), (2.0 ** -102, True), (np.nextafter(np.float32(2.0 ** -102), np.float32(0)), False), (np.nextafter(np.float32(2.0 ** 96), np.float32(0)), True),
                      (2.0 ** 96, False), (-0.0, False), (1e-45, False), (np.inf, False)]:
        m = _const_tree_model([value] + [0.0] * 7, 4, 8, 1)
        for sum_mode in (0, 2, 1):
            e = _engine(mock)
            p = _params(m, sum_mode)
            rc = mock.ddt_load_model(e, C.byref(p), m.wlines.ctypes.data, m.wlines.size // 4, m.flines.ctypes.data, m.flines.size // 8)
            assert rc == (0 if ok or sum_mode == 1 else -5), (value, sum_mode, rc)
            mock.ddt_destroy(e)


# ------------------------------------------------------------------------------------------------- GPU: the kernels
def _gpu_variants(eng, m, sum_mode):
    ids = []
    for v, _name in enumerate(ddt.variant_names()):
        try:
            eng.set_option("variant", v)
            eng.load_model(_params(m, sum_mode), m.wlines, m.flines)
            ids.append(v)
        except ddt.DDTError as ex:
            assert ex.code == -5
    eng.set_option("variant", -1)
    return ids


@pytest.mark.gpu
@pytest.mark.parametrize("where,D,F", [(w, 8, 32) for w in WHERE[:-1]] + [("tree8", 6, 28), ("slot", 4, 16), ("cluster", 11, 40)])
def test_gpu_crafted_every_variant(where, D, F):
    m, _ = _crafted(where, D, F)
    x = O.gen_tuples(0, 1100, F)
    e = ddt.Engine(0)
    names = ddt.variant_names()
    vids = _gpu_variants(e, m, 0)
    assert 0 in vids and (len(vids) >= 2 or D > 10)
    for v in vids:
        for sum_mode, want in ((0, IEEE_BITS), (2, RTL_BITS)):
            e.set_option("variant", v)
            e.load_model(_params(m, sum_mode), m.wlines, m.flines)
            got = e.score(x)
            assert set(_bits(got)) == {want}, (names[v], sum_mode, hex(int(_bits(got)[0])))
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("T,D,F,seed", [(64, 3, 8, 0), (200, 6, 28, 1), (300, 8, 32, 2), (96, 8, 64, 3), (24, 10, 32, 4)])
def test_gpu_corner_rich_every_variant(T, D, F, seed):
    m = _corner_rich(T, D, F, seed)
    x = O.gen_tuples(0, 2500, F)
    rtl, ieee = O.score(m, x, sum_mode=O.SUM_REF_FLOPOCO), O.score(m, x, sum_mode=O.SUM_REF_NATIVE)
    assert (_bits(rtl) != _bits(ieee)).mean() > 0.02
    e = ddt.Engine(0)
    names = ddt.variant_names()
    for v in _gpu_variants(e, m, 0):
        for sum_mode, want in ((0, ieee), (2, rtl)):
            e.set_option("variant", v)
            e.load_model(_params(m, sum_mode), m.wlines, m.flines)
            got = e.score(x)
            assert np.array_equal(_bits(got), _bits(want)), (names[v], sum_mode, int((_bits(got) != _bits(want)).sum()))
    e.close()


@pytest.mark.gpu
def test_gpu_hop_adder():
    """Two tree shards scored separately, partials combined by the chain-sum kernel (the inter-device hop adders)."""
    import torch
    m, nd = _crafted("hop")
    x = O.gen_tuples(0, 1100, 32)
    for sum_mode, want_bits, omode in ((0, IEEE_BITS, O.SUM_REF_NATIVE), (2, RTL_BITS, O.SUM_REF_FLOPOCO)):
        want = O.score(m, x, sum_mode=omode, n_devices=nd)
        assert set(_bits(want)) == {want_bits}
        parts = []
        for g in range(nd):
            e = ddt.Engine(0)
            e.load_model(_params(m, sum_mode), m.wlines, m.flines, g, nd)
            parts.append(torch.from_numpy(e.score(x)).cuda())
            if g + 1 < nd:
                e.close()
        got = e.chain_sum_device(torch.stack(parts))
        torch.cuda.synchronize()
        assert np.array_equal(_bits(got.cpu().numpy()), _bits(want)), sum_mode
        e.close()


@pytest.mark.gpu
def test_gpu_sparse_forest_corner():
    """The sparse kernel sums through the same adder network."""
    m = _corner_rich(40, 5, 12, 7)
    p = m.params
    sp = O.sparse_from_perfect(m)
    x = O.gen_tuples(0, 2000, 12)
    e = ddt.Engine(0)
    for sum_mode, omode in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        want = O.score_sparse(sp, x, sum_mode=omode)
        e.load_model_sparse(ddt.make_sparse_params(p.num_trees, p.num_levels, p.num_features, clusters=p.clusters_per_tuple, sum_mode=sum_mode), sp.node_lines, sp.first)
        got = e.score(x)
        assert np.array_equal(_bits(got), _bits(want)), sum_mode
    e.close()


# ---- the oracle's cache-blocked scorer (orc_score_fast_ex: what bench.py checks millions of rows of a multi-rank job or of sum_mode 2
# against) is held to the plain oracle -- and its adder shortcut to the bit-level model of the FloPoCo adder -- exactly where they could part
def test_fast_oracle_adder_shortcut_equals_the_bit_level_model():
    for seed in (1, 2, 3):
        assert O.fast_add_selftest(seed, 4_000_000) == 0
    # the shortcut is exercised, not bypassed: the documented case and its mirror images through the batch scorer
    for where in WHERE:
        m, nd = _crafted(where, D=4, F=8)
        x = O.gen_tuples(1, 64, 8, 0)
        want = O.score(m, x, sum_mode=O.SUM_REF_FLOPOCO, n_devices=nd)
        assert _bits(want)[0] == RTL_BITS
        assert np.array_equal(_bits(O.score_fast(m, x, sum_mode=O.SUM_REF_FLOPOCO, n_devices=nd)), _bits(want))
        assert _bits(O.score_fast(m, x, sum_mode=O.SUM_REF_NATIVE, n_devices=nd))[0] == IEEE_BITS


@pytest.mark.parametrize("T,D,F,seed", [(64, 5, 12, 1), (100, 6, 28, 2), (250, 4, 16, 3)])
def test_fast_oracle_equals_plain_oracle_on_corner_rich_models(T, D, F, seed):
    m = _corner_rich(T, D, F, seed)
    x = O.gen_tuples(seed, 4000, F, 0)
    differ = 0
    for nd in (1, 2, 4, 8):
        for sm in (O.SUM_REF_NATIVE, O.SUM_REF_FLOPOCO):
            assert np.array_equal(_bits(O.score_fast(m, x, sum_mode=sm, n_devices=nd)), _bits(O.score(m, x, sum_mode=sm, n_devices=nd))), (nd, sm)
        differ += int((_bits(O.score(m, x, sum_mode=O.SUM_REF_NATIVE, n_devices=nd)) != _bits(O.score(m, x, sum_mode=O.SUM_REF_FLOPOCO, n_devices=nd))).sum())
    assert differ > 0   # the two adders really part on these models
    for K, inter in ((4, True), (2, False)):
        for nd in (1, 3):
            for sm in (O.SUM_REF_NATIVE, O.SUM_REF_FLOPOCO):
                la, ca = O.classify(m, x[:1500], K, inter, sum_mode=sm, n_devices=nd)
                lb, cb = O.classify_fast(m, x[:1500], K, inter, sum_mode=sm, n_devices=nd)
                assert np.array_equal(la, lb) and np.array_equal(_bits(ca), _bits(cb)), (K, inter, nd, sm)
