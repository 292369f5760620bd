"""Seeded random shapes through the engine's OWN kernel choice (and one forced alternative), every row compared
with the oracle: tree counts that are not multiples of 8 / of the chunk size, feature counts that are not
multiples of 4, ragged batch sizes, both compare modes, all cluster counts, missing values and negatives."""
import numpy as np
import pytest

from oracle import oracle as O
import ddt

pytestmark = pytest.mark.gpu


def _cases():
    rng = np.random.default_rng(20260921)
    out = []
    for i in range(44):
        D = int(rng.choice([4, 6, 8, 8, 8, 5, 7, 3, 9, 10]))
        T = int(rng.integers(1, 330)) if D <= 8 else int(rng.integers(1, 130))
        F = int(rng.integers(1, 41)) if D in (4, 6, 8, 9, 10) else int(rng.integers(1, 80))
        n = int(rng.integers(1, 7000))
        out.append((i, T, D, F, n, int(rng.integers(0, 2)), int(rng.choice([1, 2, 4, 8])), int(rng.integers(0, 2))))
    # round 5: deep perfect trees (depth 11..15 on the deep kernels, tuples of up to 64 words on their wide forms, beyond that the generic
    # kernel), a stream of their own so that the cases above stay what they were
    rng = np.random.default_rng(20260923)
    for i in range(44, 60):
        D = int(rng.choice([11, 12, 12, 13, 14, 15]))
        T = int(rng.integers(1, 40)) if D <= 13 else int(rng.integers(1, 12))
        F = int(rng.integers(1, 70))
        n = int(rng.integers(1, 5000))
        out.append((i, T, D, F, n, int(rng.integers(0, 2)), int(rng.choice([1, 2, 4, 8])), int(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("seed,T,D,F,n,cmp_mode,clusters,sum_mode", _cases())
def test_random_shape_bit_exact(seed, T, D, F, n, cmp_mode, clusters, sum_mode):
    m = O.gen_model(T, D, F, dist=1, cmp_mode=cmp_mode, clusters=clusters)
    x = O.gen_tuples(1000 + seed, n, F, dist=1, missing_bits=m.params.missing_bits)
    want = O.score(m, x, sum_mode=O.SUM_REF_FLOPOCO if sum_mode == 0 else O.SUM_F64_SEQ)
    p = m.params
    params = ddt.make_params(p.num_trees, p.num_levels, p.num_features, p.missing_bits, p.cmp_mode, p.clusters_per_tuple, sum_mode)
    e = ddt.Engine(0)
    try:
        e.load_model(params, m.wlines, m.flines)
        auto = e.info().variant_name.decode()
        got = e.score(x)
        bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
        assert bad.size == 0, f"{auto}: {bad.size} rows differ, first {bad[:5]}"
        # one forced alternative: the rank-quantised kernel where it fits, else the generic kernel
        names = ddt.variant_names()
        alt = [v for v, nm in enumerate(names) if nm.startswith("q16_d%d" % D) and nm != auto]
        forced = alt[0] if alt else 0
        try:
            e.set_option("variant", forced)  # re-packs the loaded model: DDT_EUNSUPPORTED if it does not fit
            e.load_model(params, m.wlines, m.flines)
        except ddt.DDTError as ex:
            assert ex.code == -5
        else:
            got2 = e.score(x)
            assert np.array_equal(got2.view(np.uint32), want.view(np.uint32)), names[forced]
    finally:
        e.close()


def _sparse_cases():
    rng = np.random.default_rng(20260922)
    out = []
    for i in range(24):
        D = int(rng.integers(1, 21))                     # depth bound; deeper than 16 is fine in the sparse format
        T = int(rng.integers(1, 70))
        F = int(rng.integers(1, 100))
        full = int(rng.integers(0, min(D, 7) + 1))
        pm = int(rng.integers(300, 900))
        n = int(rng.integers(1, 3000))
        out.append((i, T, D, F, full, pm, n, int(rng.integers(0, 2)), int(rng.choice([1, 2, 4, 8])), int(rng.integers(0, 2))))
    return out


@pytest.mark.parametrize("seed,T,D,F,full,pm,n,cmp_mode,clusters,sum_mode", _sparse_cases())
def test_random_sparse_forest_bit_exact(seed, T, D, F, full, pm, n, cmp_mode, clusters, sum_mode):
    """Seeded random SPARSE forests (ragged depths 1..20, tree counts that are not multiples of 8, 1..99 features, both
    compare modes, all cluster counts, missing values) through the engine's own kernel choice and the host feeder."""
    s = O.gen_sparse_model(T, D, F, full, pm, 1, cmp_mode=cmp_mode, clusters=clusters)
    x = O.gen_tuples(2000 + seed, n, F, dist=1)
    want = O.score_sparse(s, x, sum_mode=O.SUM_REF_FLOPOCO if sum_mode == 0 else O.SUM_F64_SEQ)
    q = s.params
    p = ddt.make_sparse_params(q.num_trees, q.num_levels, q.num_features, q.missing_bits, q.cmp_mode, q.clusters_per_tuple, sum_mode)
    e = ddt.Engine(0)
    try:
        e.load_model_sparse(p, s.node_lines, s.first)
        got = e.score(x)
        bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
        assert bad.size == 0, f"{e.info().variant_name.decode()}: {bad.size} rows differ, first {bad[:5]}"
        e.set_option("sparse_deep_order", 1)
        e.set_option("sparse_top_levels", 6)
        e.load_model_sparse(p, s.node_lines, s.first, 0, 1)
        assert np.array_equal(e.score(x).view(np.uint32), want.view(np.uint32))
    finally:
        e.close()


def _dist_cases():
    rng = np.random.default_rng(20260923)
    return [(i, int(rng.integers(80, 420)), int(rng.integers(1, 33)), int(rng.integers(1, 5000)), int(rng.integers(0, 2))) for i in range(14)]


@pytest.mark.parametrize("seed,T,F,n,cmp_mode", _dist_cases())
def test_rank_prepass_on_mixed_threshold_distributions(seed, T, F, n, cmp_mode):
    """The LDS-resident rank pre-pass (segmented bucket index) on models whose thresholds follow a DIFFERENT distribution per
    feature -- uniform, exponential over 30 octaves, small integers, one constant, negative, +-huge, two tight clusters --
    with tuples drawn around the thresholds (many exactly on one), missing values, both compare modes; every number of
    feature groups that fits, and the transpose + rank kernels."""
    rng = np.random.default_rng(777 + seed)
    D, nint = 8, 255
    m = O.gen_model(T, D, F, dist=1, cmp_mode=cmp_mode)
    w = m.wlines.copy().reshape(T, -1)
    fidx = rng.integers(0, F, (T, nint))
    kinds = rng.integers(0, 7, F)
    thr = np.empty((T, nint), np.float32)
    for j in range(F):
        sel = fidx == j
        k = int(sel.sum())
        if k == 0:
            continue
        kind = kinds[j]
        if kind == 0:
            v = rng.random(k)
        elif kind == 1:
            v = np.exp2(rng.uniform(-20, 10, k))
        elif kind == 2:
            v = rng.integers(-5, 40, k).astype(np.float64)
        elif kind == 3:
            v = np.full(k, 0.75)
        elif kind == 4:
            v = -np.exp2(rng.uniform(-3, 3, k))
        elif kind == 5:
            v = rng.choice([-3.0e38, 3.0e38, 1e-30, -1e-30, 0.0, 1.0], k)
        else:
            v = np.where(rng.random(k) < 0.5, 0.25 + rng.integers(0, 4000, k) * 2.0 ** -24, 1000.0 + rng.integers(0, 4000, k) * 2.0 ** -12)
        thr[sel] = v.astype(np.float32)
    w[:, :nint] = thr.view(np.uint32)
    # feature-index lines: one u16 entry per node, keep the flag bits of the generated model, replace the index
    fl = m.flines.copy().reshape(T, -1)
    fl[:, :nint] = (fl[:, :nint] & 0xF800) | fidx.astype(fl.dtype)
    m = O.Model(m.params, w.reshape(m.wlines.shape), fl.reshape(m.flines.shape))
    x = O.gen_tuples(9000 + seed, n, F, dist=1, missing_bits=m.params.missing_bits)
    pick = thr.reshape(-1)[rng.integers(0, thr.size, (n, F))].view(np.uint32).astype(np.int64) + rng.integers(-1, 2, (n, F))
    keep = rng.random((n, F)) < 0.6
    x[:, :F] = np.where(keep, (pick & 0xFFFFFFFF).astype(np.uint32), x[:, :F])
    want = O.score(m, x)
    p = m.params
    params = ddt.make_params(p.num_trees, p.num_levels, p.num_features, p.missing_bits, p.cmp_mode, p.clusters_per_tuple, 0)
    e = ddt.Engine(0)
    try:
        e.set_option("variant", ddt.variant_names().index("q16_d8_c8_u4_gl"))
        for groups in (0, 1, 2, 4, 8, -1):
            e.set_option("q16_fused_prepass", 0 if groups < 0 else 1)
            e.set_option("q16_grouped_prepass", 0 if groups < 0 else 1)
            e.set_option("q16_prepass_groups", max(groups, 0))
            e.load_model(params, m.wlines, m.flines)
            got = e.score(x)
            bad = np.flatnonzero(got.view(np.uint32) != want.view(np.uint32))
            assert bad.size == 0, f"groups={groups} (plan {e.info().prepass_groups}): {bad.size} rows differ, first {bad[:5]}"
    finally:
        e.close()
