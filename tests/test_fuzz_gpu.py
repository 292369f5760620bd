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
