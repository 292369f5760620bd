"""The Python plumbing above the C-ABI (ddt/engine.py Engine / Group, ddt/importer.py) on the CPU model of the host side
(tests/test_engine_mock.py: the real csrc/*.cpp built against tests/mock_hip/).  The test binds the ctypes layer to that build
for the duration of a test (no switch exists in the package itself: ddt._lib.lib() only ever loads libddt.so) and runs what
the GPU tests run through host buffers: scikit-learn models imported, loaded, scored and classified, sparse forests, the
single-process multi-GPU group."""
import numpy as np
import pytest

import ddt
from ddt import _lib, importer as I
from oracle import oracle as O
from tests.test_engine_mock import _build

sklearn = pytest.importorskip("sklearn")
from sklearn import ensemble  # noqa: E402


@pytest.fixture()
def on_model(monkeypatch):
    L = _build("libddt_host_mock.so")
    L.mock_reset(2, 3, 8)
    monkeypatch.setattr(_lib, "_lib", L)
    yield L


def _data(n, F, seed):
    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, F)).astype(np.float32)
    y = (X[:, 0] * 2 - X[:, 1] ** 2 + np.sin(X[:, 2] * 3) + rng.normal(scale=0.1, size=n)).astype(np.float32)
    return X, y


def test_imported_sklearn_models_through_engine(on_model):
    X, y = _data(1500, 12, 11)
    Xt = _data(1200, 12, 12)[0]
    tl = O.tuples_from_float(Xt)
    e = ddt.Engine(0)
    rf = ensemble.RandomForestRegressor(n_estimators=40, max_depth=8, random_state=0).fit(X, y)
    im = I.from_sklearn(rf)
    for sum_mode in (0, 1):
        e.load_model(im.params(sum_mode=sum_mode), im.wlines, im.flines)
        got = e.score(tl)
        assert np.allclose(got, rf.predict(Xt), rtol=2e-5, atol=2e-5)
        om = O.Model(O.make_params(im.num_trees, im.num_levels, im.num_features, im.missing_bits, im.cmp_mode, im.params().clusters_per_tuple),
                     im.wlines, im.flines)                          # the imported model uses the IEEE comparator (negative features)
        assert np.array_equal(got.view(np.uint32), O.score(om, tl, sum_mode=O.SUM_REF_FLOPOCO if sum_mode == 0 else O.SUM_F64_SEQ).view(np.uint32))
    assert e.info().tree_end - e.info().tree_begin == im.num_trees and e.stats().tuples_out == 2 * len(Xt)
    lab = np.digitize(y, np.quantile(y, [0.33, 0.66]))
    gbc = ensemble.GradientBoostingClassifier(n_estimators=12, max_depth=4, random_state=0).fit(X, lab)
    im = I.from_sklearn(gbc)
    e.load_model_multiclass(im.params(sum_mode=1), im.wlines, im.flines, im.num_classes, True)
    labels, cs = e.classify(tl, want_scores=True)
    raw = cs.T.astype(np.float64) + im.base_score[None, :]
    assert np.allclose(raw, gbc.decision_function(Xt), rtol=1e-4, atol=1e-4)
    assert np.mean(np.argmax(raw, axis=1) == gbc.predict(Xt)) > 0.995
    assert np.array_equal(labels, np.argmax(cs, axis=0))            # the engine's argmax is over the raw class sums (no prior added)
    with pytest.raises(ddt.DDTError):
        e.score(tl)                                                  # the scalar call refuses a multi-class model
    e.close()


def test_imported_deep_forest_as_a_sparse_stream(on_model):
    X, y = _data(2500, 20, 5)
    Xt = _data(900, 20, 6)[0]
    tl = O.tuples_from_float(Xt)
    rf = ensemble.RandomForestRegressor(n_estimators=24, max_depth=14, random_state=1).fit(X, y)
    im = I.from_sklearn(rf, sparse=True)
    e = ddt.Engine(0)
    im.load_into(e)
    got = e.score(tl)
    assert np.allclose(got, rf.predict(Xt), rtol=2e-5, atol=2e-5)
    assert e.info().variant_name.decode().startswith("sparse_")
    e.close()


def test_group_of_four_devices(on_model):
    T, D, F, n = 200, 8, 32, 3000
    m, x = O.gen_model(T, D, F, 0), O.gen_tuples(0, n, F, 0)
    g = ddt.Group([0, 1, 2, 3])
    g.load_model(ddt.make_params(T, D, F), m.wlines, m.flines)
    for combine in (ddt.COMBINE_ALLREDUCE, ddt.COMBINE_CHAIN):
        assert np.array_equal(g.score(x, combine=combine).view(np.uint32), O.score(m, x, n_devices=4).view(np.uint32))
    with pytest.raises(ddt.DDTError):
        g.score_rows(x)                                              # tree shards loaded: not the row mode
    g.load_model_replicated(ddt.make_params(T, D, F), m.wlines, m.flines)
    assert np.array_equal(g.score_rows(x).view(np.uint32), O.score(m, x).view(np.uint32))
    g.close()
    with pytest.raises(ddt.DDTError):
        ddt.Group([0, 0])                                             # one communicator rank per device


def test_hybrid_through_the_python_bindings(on_model):
    """ddt.Group(tree_ranks=) and ddt.Comm(tree_ranks=) (ddt_group_create_hybrid / ddt_comm_create_hybrid): four "devices" as 2 row groups x 2
    tree shards -- the single-process group, and one thread per rank calling the host-buffer form (ncclCommSplit behind the C-ABI)."""
    import threading

    T, D, F, n = 200, 8, 32, 5000
    m, x = O.gen_model(T, D, F, 0), O.gen_tuples(0, n, F, 0)
    want = O.score(m, x, n_devices=2)
    g = ddt.Group([0, 1, 2, 3], tree_ranks=2)
    g.load_model(ddt.make_params(T, D, F), m.wlines, m.flines)
    assert np.array_equal(g.score(x, combine=ddt.COMBINE_CHAIN).view(np.uint32), want.view(np.uint32))
    g.close()
    assert ddt.hybrid_rows(n, 2, 0) == (0, 3072) and ddt.hybrid_rows(n, 2, 1) == (3072, n)
    uid, got, errs = ddt.comm_unique_id(), {}, []

    def rank(r):
        try:
            on_model.hipSetDevice(r)
            e = ddt.Engine(r)
            e.load_model(ddt.make_params(T, D, F), m.wlines, m.flines, r % 2, 2)
            c = ddt.Comm(e, r, 4, uid, tree_ranks=2)
            lay = c.layout()
            assert (lay.rank, lay.n_ranks, lay.tree_ranks, lay.tree_rank, lay.row_groups, lay.row_group) == (r, 4, 2, r % 2, 2, r // 2)
            c.set_option("host_rows", 2100)
            got[r] = c.score(x, combine=ddt.COMBINE_CHAIN)
            done.wait()
            c.close()
            e.close()
        except BaseException as ex:  # noqa: BLE001
            errs.append((r, repr(ex)))
            done.abort()

    done = threading.Barrier(4)
    th = [threading.Thread(target=rank, args=(r,)) for r in range(4)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errs, errs
    for r in range(4):
        assert np.array_equal(got[r].view(np.uint32), want.view(np.uint32)), r


def test_comm_host_buffer_call(on_model):
    T, D, F, n = 90, 8, 32, 2100
    m, x = O.gen_model(T, D, F, 1), O.gen_tuples(0, n, F, 1)
    e = ddt.Engine(0)
    e.load_model(ddt.make_params(T, D, F), m.wlines, m.flines)
    c = ddt.Comm(e, 0, 1, ddt.comm_unique_id())
    c.set_option("host_rows", 800)
    for combine in (ddt.COMBINE_ALLREDUCE, ddt.COMBINE_CHAIN):
        assert np.array_equal(c.score(x, combine=combine).view(np.uint32), O.score(m, x).view(np.uint32))
    with pytest.raises(ddt.DDTError):
        c.set_option("tuple_broadcast", 2)
    c.close()
    e.close()
