"""the model importer (ddt/importer.py).  CPU tests score the imported streams with the oracle,
the GPU test with the engine; both must reproduce scikit-learn's own predictions."""
import numpy as np
import pytest

from oracle import oracle as O
import ddt
from ddt import importer as I

sk = pytest.importorskip("sklearn")
from sklearn import ensemble, tree  # noqa: E402


def _omodel(im, clusters=None):
    p = O.make_params(im.num_trees, im.num_levels, im.num_features, im.missing_bits, im.cmp_mode, clusters)
    return O.Model(p, im.wlines, im.flines)


def _data(n=1200, F=9, seed=0):
    rng = np.random.default_rng(seed)
    X = rng.standard_normal((n, F)).astype(np.float32)  # negatives on purpose: needs the IEEE comparator
    y = (np.sin(2 * X[:, 0]) + X[:, 1] * X[:, 2] - 0.5 * X[:, 3] + 0.1 * rng.standard_normal(n)).astype(np.float32)
    return X, y


def test_le_to_lt_threshold_is_exact():
    rng = np.random.default_rng(1)
    t = np.concatenate([rng.standard_normal(2000), rng.standard_normal(2000).astype(np.float32).astype(np.float64), [0.0, -0.0, 1e-40, -1e-40]])
    thr = I.le_to_lt_threshold(t)
    for probe in (t.astype(np.float32), np.nextafter(t.astype(np.float32), np.float32(np.inf)), np.nextafter(t.astype(np.float32), np.float32(-np.inf))):
        assert np.array_equal(probe.astype(np.float64) <= t, probe < thr)


@pytest.mark.parametrize("maker", [
    lambda: tree.DecisionTreeRegressor(max_depth=6, random_state=0),
    lambda: ensemble.RandomForestRegressor(n_estimators=25, max_depth=7, random_state=0),
    lambda: ensemble.ExtraTreesRegressor(n_estimators=10, max_depth=5, random_state=0),
    lambda: ensemble.GradientBoostingRegressor(n_estimators=40, max_depth=3, random_state=0),
])
def test_sklearn_regressors_through_oracle(maker):
    X, y = _data()
    mdl = maker().fit(X, y)
    im = I.from_sklearn(mdl)
    Xt = _data(500, seed=5)[0]
    got = O.score(_omodel(im), O.tuples_from_float(Xt), sum_mode=O.SUM_F64_SEQ) + np.float32(im.base_score[0])
    assert np.allclose(got, mdl.predict(Xt), rtol=2e-5, atol=2e-5)
    # per-tree exactness: every imported tree picks the leaf sklearn picks
    if hasattr(mdl, "estimators_") and not isinstance(mdl, ensemble.GradientBoostingRegressor):
        lv = O.leaves(_omodel(im), O.tuples_from_float(Xt)[7]).view(np.float32) * len(mdl.estimators_)
        want = np.array([e.predict(Xt[7:8])[0] for e in mdl.estimators_])
        assert np.allclose(lv, want, rtol=1e-6, atol=1e-7)


def test_sklearn_missing_values_follow_default_direction():
    X, y = _data(1500, 6, 3)
    Xm = X.copy()
    Xm[np.random.default_rng(2).random(X.shape) < 0.1] = np.nan
    mdl = tree.DecisionTreeRegressor(max_depth=6, random_state=0).fit(Xm, y)
    im = I.from_sklearn(mdl)
    Xt = _data(400, 6, 9)[0]
    Xt[np.random.default_rng(4).random(Xt.shape) < 0.15] = np.nan
    tl = O.tuples_from_float(Xt)
    assert (tl[:, :6] == 0x7FC00000).sum() > 100  # np.nan is the canonical quiet NaN the engine treats as missing
    got = O.score(_omodel(im), tl, sum_mode=O.SUM_F64_SEQ)
    assert np.allclose(got, mdl.predict(Xt), rtol=1e-6, atol=1e-6)


def test_sklearn_classifiers_through_oracle():
    X, y = _data(2000, 8, 7)
    lab = np.digitize(y, np.quantile(y, [0.25, 0.5, 0.75]))  # 4 classes
    Xt = _data(600, 8, 8)[0]
    gbc = ensemble.GradientBoostingClassifier(n_estimators=15, max_depth=3, random_state=0).fit(X, lab)
    im = I.from_sklearn(gbc)
    assert im.num_classes == 4 and im.num_trees == 60
    _, cs = O.classify(_omodel(im), O.tuples_from_float(Xt), 4, sum_mode=O.SUM_F64_SEQ)
    raw = cs.T.astype(np.float64) + im.base_score[None, :]
    assert np.allclose(raw, gbc.decision_function(Xt), rtol=1e-4, atol=1e-4)
    assert np.mean(np.argmax(raw, axis=1) == gbc.predict(Xt)) > 0.995
    rfc = ensemble.RandomForestClassifier(n_estimators=12, max_depth=6, random_state=0).fit(X, lab)
    im = I.from_sklearn(rfc)
    labels, cs = O.classify(_omodel(im), O.tuples_from_float(Xt), 4, sum_mode=O.SUM_F64_SEQ)
    assert np.allclose(cs.T, rfc.predict_proba(Xt), atol=1e-5)
    assert np.mean(labels == rfc.predict(Xt)) > 0.99


def test_sklearn_hist_gradient_boosting_through_oracle():
    """HistGradientBoosting (binned training, real-valued thresholds at predict time, native missing-value support):
    raw predictions of regressor, binary and multi-class classifier reproduced through the reference format."""
    X, y = _data(3000, 10, 11)
    Xm = X.copy()
    Xm[np.random.default_rng(3).random(X.shape) < 0.05] = np.nan
    Xt = _data(700, 10, 12)[0]
    Xt[np.random.default_rng(5).random(Xt.shape) < 0.08] = np.nan
    tl = O.tuples_from_float(Xt)
    reg = ensemble.HistGradientBoostingRegressor(max_iter=30, max_depth=6, random_state=0).fit(Xm, y)
    im = I.from_sklearn(reg)
    assert im.num_classes == 1 and im.num_trees == 30 and im.num_levels <= 6
    got = O.score(_omodel(im), tl, sum_mode=O.SUM_F64_SEQ).astype(np.float64) + im.base_score[0]
    assert np.allclose(got, reg.predict(Xt), rtol=2e-5, atol=2e-5)
    lab = np.digitize(y, np.quantile(y, [1 / 3, 2 / 3]))  # 3 classes
    clf = ensemble.HistGradientBoostingClassifier(max_iter=12, max_depth=5, random_state=0).fit(Xm, lab)
    im = I.from_sklearn(clf)
    assert im.num_classes == 3 and im.num_trees == 36
    labels, cs = O.classify(_omodel(im), tl, 3, sum_mode=O.SUM_F64_SEQ)
    raw = cs.T.astype(np.float64) + im.base_score[None, :]
    assert np.allclose(raw, clf.decision_function(Xt), rtol=1e-4, atol=1e-4)
    assert np.mean(np.argmax(raw, axis=1) == clf.predict(Xt)) > 0.995
    binc = ensemble.HistGradientBoostingClassifier(max_iter=10, max_depth=4, random_state=0).fit(Xm, (y > np.median(y)).astype(int))
    im = I.from_sklearn(binc)
    assert im.num_classes == 1
    got = O.score(_omodel(im), tl, sum_mode=O.SUM_F64_SEQ).astype(np.float64) + im.base_score[0]
    assert np.allclose(got, binc.decision_function(Xt), rtol=1e-4, atol=1e-4)


def _xgb_eval(tr, x):
    n = 0
    while tr["left_children"][n] >= 0:
        f, c = tr["split_indices"][n], np.float32(tr["split_conditions"][n])
        v = x[f]
        go_left = bool(tr["default_left"][n]) if np.isnan(v) else bool(v < c)
        n = tr["left_children"][n] if go_left else tr["right_children"][n]
    return tr["split_conditions"][n]


def test_xgboost_json_dump():
    t0 = {"left_children": [1, 3, -1, -1, -1], "right_children": [2, 4, -1, -1, -1], "split_indices": [0, 1, 0, 0, 0],
          "split_conditions": [0.5, -0.25, 0.3, -0.1, 0.2], "default_left": [1, 0, 0, 0, 0]}
    t1 = {"left_children": [1, -1, -1], "right_children": [2, -1, -1], "split_indices": [2, 0, 0],
          "split_conditions": [0.0, 1.5, -1.5], "default_left": [0, 0, 0]}
    j = {"learner": {"learner_model_param": {"num_feature": "3", "num_class": "0", "base_score": "0.5"},
                     "gradient_booster": {"name": "gbtree", "model": {"trees": [t0, t1], "tree_info": [0, 0]}}}}
    im = I.from_xgboost_json(j)
    assert (im.num_trees, im.num_levels, im.num_features) == (2, 2, 3)
    rng = np.random.default_rng(0)
    X = rng.standard_normal((300, 3)).astype(np.float32)
    X[rng.random(X.shape) < 0.2] = np.nan
    X[0] = [0.5, -0.25, 0.0]  # exactly on the thresholds: '<' sends them right
    got = O.score(_omodel(im), O.tuples_from_float(X), sum_mode=O.SUM_F64_SEQ)
    want = np.array([_xgb_eval(t0, x) + _xgb_eval(t1, x) for x in X], np.float32)
    assert np.allclose(got, want, atol=1e-7)


def test_leaf_values_land_in_the_loaders_exact_domain():
    """The importer's leaf rule == the loader's check (csrc/ddt_model.cpp leaf_outside_exact_domain: +0 or a normal with
    2^-102 <= |v| < 2^96): what it emits loads with the default options; what it cannot represent raises instead of failing later."""
    for v, want in ((1e-35, 0.0), (-1e-35, 0.0), (-0.0, 0.0), (1e-45, 0.0), (2.0 ** -102, 2.0 ** -102), (-(2.0 ** -102), -(2.0 ** -102)),
                    (0.1, np.float32(0.1)), (2.0 ** 95, 2.0 ** 95)):
        got = I.leaf_f32(v)
        assert got == np.float32(want) and not np.signbit(got) or got == np.float32(want) and want < 0, v
        bits = int(np.float32(got).view(np.uint32))
        ex = (bits >> 23) & 0xFF
        assert bits == 0 or 25 <= ex <= 222, (v, hex(bits))               # the loader's test, restated
    for v in (2.0 ** 96, -1e30, np.inf, -np.inf, np.nan, 1e39):
        with pytest.raises(ValueError):
            I.leaf_f32(v)
    # through a model: a tiny leaf is flushed (the model still loads), a huge one is refused at import time
    t = {"left_children": [1, -1, -1], "right_children": [2, -1, -1], "split_indices": [0, 0, 0], "split_conditions": [0.5, 1e-35, 0.25], "default_left": [0, 0, 0]}
    j = {"learner": {"learner_model_param": {"num_feature": "1", "num_class": "0", "base_score": "0"},
                     "gradient_booster": {"name": "gbtree", "model": {"trees": [t], "tree_info": [0]}}}}
    im = I.from_xgboost_json(j)
    assert sorted(np.unique(im.wlines.view(np.float32)[1:3])) == [0.0, 0.25]
    import ctypes as C

    L, info = ddt.lib(), (C.c_uint64 * 12)()
    w, f = np.ascontiguousarray(im.wlines).view(np.uint32), np.ascontiguousarray(im.flines).view(np.uint16)
    rc = L.ddt_debug_model_image(C.byref(im.params()), w.ctypes.data, w.size // 4, f.ctypes.data, f.size // 8, -1, None, None, 0, None, 0, C.byref(info))
    assert rc == 0, rc                                                      # the loader's own validation accepts the stream (host-only hook)
    w2 = w.copy()
    w2[np.flatnonzero(w2.view(np.float32) == np.float32(0.25))[0]] = np.float32(1e-35).view(np.uint32)
    assert L.ddt_debug_model_image(C.byref(im.params()), w2.ctypes.data, w2.size // 4, f.ctypes.data, f.size // 8, -1, None, None, 0, None, 0, C.byref(info)) == -5
    t["split_conditions"][1] = 1e30
    with pytest.raises(ValueError):
        I.from_xgboost_json(j)


@pytest.mark.gpu
def test_imported_models_on_gpu():
    X, y = _data(3000, 12, 11)
    Xt = _data(5000, 12, 12)[0]
    tl = O.tuples_from_float(Xt)
    e = ddt.Engine(0)
    rf = ensemble.RandomForestRegressor(n_estimators=64, max_depth=8, random_state=0).fit(X, y)
    im = I.from_sklearn(rf)
    for sum_mode in (0, 1):
        e.load_model(im.params(sum_mode=sum_mode), im.wlines, im.flines)
        got = e.score(tl)
        assert np.allclose(got, rf.predict(Xt), rtol=2e-5, atol=2e-5)
        om = _omodel(im, im.params().clusters_per_tuple)
        assert np.array_equal(got.view(np.uint32), O.score(om, tl, sum_mode=O.SUM_REF_FLOPOCO if sum_mode == 0 else O.SUM_F64_SEQ).view(np.uint32))
    # a DEEP forest in the perfect format (early leaves padded down to depth 11 / 12): the deep rank-quantised kernels (round 5)
    for depth, name in ((11, "q16d_d11_k8_c8_u4_cm"), (12, "q16d_d12_k9_c4_u4_cm")):
        rfd = ensemble.RandomForestRegressor(n_estimators=24, max_depth=depth, random_state=1).fit(X, y)
        im = I.from_sklearn(rfd)
        assert im.num_levels == depth
        e.load_model(im.params(), im.wlines, im.flines)
        assert e.info().variant_name.decode() == name and e.info().fallback_kernel == 0
        got = e.score(tl)
        assert np.allclose(got, rfd.predict(Xt), rtol=2e-5, atol=2e-5)
        om = _omodel(im, im.params().clusters_per_tuple)
        assert np.array_equal(got.view(np.uint32), O.score_fast(om, tl).view(np.uint32))
    lab = np.digitize(y, np.quantile(y, [0.33, 0.66]))
    gbc = ensemble.GradientBoostingClassifier(n_estimators=20, max_depth=4, random_state=0).fit(X, lab)
    im = I.from_sklearn(gbc)
    e.load_model_multiclass(im.params(sum_mode=1), im.wlines, im.flines, im.num_classes, True)
    labels, cs = e.classify(tl, want_scores=True)
    raw = cs.T.astype(np.float64) + im.base_score[None, :]
    assert np.allclose(raw, gbc.decision_function(Xt), rtol=1e-4, atol=1e-4)
    assert np.mean(np.argmax(raw, axis=1) == gbc.predict(Xt)) > 0.995
    e.close()


def _xgb_json(objective, base_score, K=0):
    """A hand-written XGBoost model dump: one stump per class, `x0 < 0.5 ? -1 : +1`."""
    tree = {"left_children": [1, -1, -1], "right_children": [2, -1, -1], "split_conditions": [0.5, -1.0, 1.0],
            "split_indices": [0, 0, 0], "default_left": [1, 0, 0]}
    n = max(1, K)
    return {"learner": {"objective": {"name": objective},
                        "learner_model_param": {"num_feature": "3", "num_class": str(K), "base_score": base_score},
                        "gradient_booster": {"name": "gbtree", "model": {"trees": [tree] * n, "tree_info": list(range(n))}}}}


def test_xgboost_base_score_goes_through_the_objectives_link():
    """learner_model_param.base_score lives in OUTPUT space: logit for the logistic objectives, log for poisson /
    gamma / tweedie, identity otherwise; XGBoost >= 2 writes it as a bracketed string."""
    im = ddt.importer.from_xgboost_json(_xgb_json("binary:logistic", "[3E-1]"))
    assert np.allclose(im.base_score, [np.log(0.3 / 0.7)])
    im = ddt.importer.from_xgboost_json(_xgb_json("binary:logistic", "0.5"))
    assert np.allclose(im.base_score, [0.0])
    im = ddt.importer.from_xgboost_json(_xgb_json("count:poisson", "2.5E0"))
    assert np.allclose(im.base_score, [np.log(2.5)])
    im = ddt.importer.from_xgboost_json(_xgb_json("reg:squarederror", "[1.25E0]"))
    assert np.allclose(im.base_score, [1.25])
    im = ddt.importer.from_xgboost_json(_xgb_json("multi:softprob", "[1E-1,2E-1,3E-1]", K=3))
    assert np.allclose(im.base_score, [0.1, 0.2, 0.3]) and im.num_classes == 3
    im = ddt.importer.from_xgboost_json(_xgb_json("multi:softprob", "5E-1", K=3))
    assert np.allclose(im.base_score, [0.5] * 3)
    with pytest.raises(TypeError):
        ddt.importer.from_xgboost_json(_xgb_json("rank:some_new_objective", "0.5"))
    with pytest.raises(ValueError):
        ddt.importer.from_xgboost_json(_xgb_json("binary:logistic", "1.5"))
    # margin = tree sum + base: the oracle on the imported streams
    im = ddt.importer.from_xgboost_json(_xgb_json("binary:logistic", "[3E-1]"))
    m = O.Model(O.make_params(1, im.num_levels, 3, cmp_mode=1), im.wlines, im.flines)
    x = O.tuples_from_float(np.array([[0.2, 0, 0], [0.7, 0, 0]], np.float32))
    assert np.allclose(O.score(m, x) + im.base_score[0], [-1 + np.log(0.3 / 0.7), 1 + np.log(0.3 / 0.7)])
    sp = ddt.importer.from_xgboost_json(_xgb_json("binary:logistic", "[3E-1]"), sparse=True)
    s = O.SparseModel(O.make_sparse_params(1, sp.num_levels, 3, cmp_mode=1), sp.node_lines, sp.tree_first_line)
    assert np.array_equal(O.score_sparse(s, x), O.score(m, x))
