"""Sparse (explicit-children) forests -- BASELINE config 4, SURVEY 8(f) N2 second half.

The reference has no such format (its engine takes perfect trees only, DTPU.sv:20-28; the hook for bigger trees --
entry bit 14 / PartialTrees -- is disabled and ill-defined), so parity is anchored like this:

  CPU (-m "not gpu"):  the sparse oracle (orc_traverse_sparse / orc_score_sparse) == pad_to_perfect
                       (orc_sparse_to_perfect) + the perfect-tree oracle, which is the one pinned to the RTL-evaluated
                       vectors; perfect -> sparse -> score agrees too; the importer's sparse stream == its padded
                       perfect stream == scikit-learn's own predictions.
  GPU (-m gpu):        the HIP sparse kernel, through the C-ABI, bit-exact against the sparse oracle: every K (levels
                       staged in LDS), both deep-record orders, missing values, both comparators, both sum modes, tree
                       shards (incl. empty ones), trees shallower than K, single-leaf trees, and a real
                       RandomForestRegressor(n_estimators=512, max_depth=16) on 64 features.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
import ddt


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


# T, max_depth, F, full_levels, split_permille, dist, rows
SHAPES = [
    (16, 12, 20, 4, 600, 1, 700),
    (40, 16, 64, 6, 700, 0, 600),     # config-4-like: 64 features, depth 16
    (9, 9, 7, 2, 800, 1, 513),        # fewer than 8 trees in the last group, F not a multiple of 4
    (24, 5, 12, 2, 500, 1, 400),      # shallower than the smallest K: no deep records at all
    (8, 14, 33, 1, 750, 1, 300),      # very ragged
    (3, 1, 4, 1, 0, 0, 100),          # depth 1: every tree is root + two leaves
]


@pytest.mark.parametrize("shape", SHAPES)
@pytest.mark.parametrize("cmp_mode", [0, 1])
def test_sparse_oracle_equals_padded_perfect_oracle(shape, cmp_mode):
    T, D, F, full, pm, dist, rows = shape
    s = O.gen_sparse_model(T, D, F, full, pm, dist, cmp_mode=cmp_mode)
    assert s.check() == 0
    x = O.gen_tuples(0, rows, F, dist)
    m = O.sparse_to_perfect(s)  # pad_to_perfect: leaves above depth D -> dummy sub-trees
    for sum_mode in (O.SUM_REF_FLOPOCO, O.SUM_REF_NATIVE, O.SUM_F64_SEQ):
        a = O.score_sparse(s, x, sum_mode=sum_mode)
        b = O.score(m, x, sum_mode=sum_mode)
        assert np.array_equal(_bits(a), _bits(b)), (shape, cmp_mode, sum_mode)
    # tree by tree, tuple by tuple: the walks select the same leaves
    for r in range(0, rows, max(1, rows // 25)):
        for t in range(T):
            assert O.traverse_sparse(s, x[r], t) == O.lib().orc_traverse(C.byref(m.params), O._p(m.wlines), O._p(m.flines), O._p(x[r]), t)
    # tree-sharded multi-device model: the same shard boundaries and chain order
    a = O.score_sparse(s, x, n_devices=min(3, T))
    b = O.score(m, x, n_devices=min(3, T))
    assert np.array_equal(_bits(a), _bits(b))


@pytest.mark.parametrize("shape", SHAPES)
def test_cpu_baseline_form_of_the_sparse_oracle_is_identical(shape):
    """orc_score_sparse_fast (bench.py --config 4 cpu_baseline: one tree at a time over a block of rows) == orc_score_sparse."""
    T, D, F, full, pm, dist, rows = shape
    s = O.gen_sparse_model(T, D, F, full, pm, dist)
    x = O.gen_tuples(7, rows + 1030, F, dist)  # more than one row block, ragged tail
    for mode in (O.SUM_REF_NATIVE, O.SUM_F64_SEQ, O.SUM_REF_FLOPOCO):
        assert np.array_equal(_bits(O.score_sparse(s, x, sum_mode=mode)), _bits(O.score_sparse_fast(s, x, sum_mode=mode))), (shape, mode)


def test_sparse_classes_oracle_equals_padded_perfect_oracle():
    """orc_classify_sparse == pad_to_perfect + orc_classify (labels and per-class sums), both class layouts, sharded too"""
    for (T, D, F, K, inter) in [(30, 10, 12, 3, True), (24, 7, 9, 4, False), (21, 12, 20, 7, True)]:
        s = O.gen_sparse_model(T, D, F, 3, 650, 1, clusters=1)
        x = O.gen_tuples(5, 600, F, 1)
        m = O.sparse_to_perfect(s)
        for nd in (1, 2):
            la, ca = O.classify_sparse(s, x, K, inter, n_devices=nd)
            lb, cb = O.classify(m, x, K, inter, n_devices=nd)
            assert np.array_equal(la, lb) and np.array_equal(_bits(ca), _bits(cb)), (T, K, inter, nd)


def test_perfect_to_sparse_roundtrip():
    m = O.gen_model(37, 6, 28, 1)
    s = O.sparse_from_perfect(m)
    assert s.check() == 0 and s.n_lines == 37 * 63
    x = O.gen_tuples(5, 900, 28, 1)
    assert np.array_equal(_bits(O.score(m, x)), _bits(O.score_sparse(s, x)))
    back = O.sparse_to_perfect(s)
    assert np.array_equal(back.wlines, m.wlines) and np.array_equal(back.flines & 0x27FF, m.flines & 0x27FF)


def test_sparse_check_rejects_malformed_streams():
    s = O.gen_sparse_model(4, 8, 10, 3, 600, 0)
    bad = O.SparseModel(s.params, s.node_lines.copy(), s.first)
    k = int(np.nonzero((bad.node_lines[:, 1] & 0x4000) == 0)[0][0])  # a node whose left child is internal
    bad.node_lines[k, 2] = 0  # child index not after its parent
    assert bad.check() == -4
    bad = O.SparseModel(s.params, s.node_lines.copy(), s.first)
    bad.node_lines[0, 1] |= 0x7FF  # feature index out of range
    assert bad.check() == -3
    shallow = O.Params(4, 3, 10, 0x7FC00000, 0, 0, 0, 1)
    assert O.SparseModel(shallow, s.node_lines, s.first).check() == -5  # deeper than num_levels


def test_library_and_oracle_generators_agree():
    for (T, D, F, full, pm, dist) in [(16, 12, 20, 4, 600, 1), (5, 16, 64, 8, 650, 0), (3, 1, 4, 1, 0, 0)]:
        lines, first = ddt.synth_sparse_model(T, D, F, full, pm, dist)
        s = O.gen_sparse_model(T, D, F, full, pm, dist)
        assert np.array_equal(lines, s.node_lines) and np.array_equal(first, s.first)


def _rf(n_estimators, max_depth, F=16, n=4000, seed=0):
    from sklearn.ensemble import RandomForestRegressor

    rng = np.random.default_rng(seed)
    X = rng.normal(size=(n, F)).astype(np.float32)
    y = (np.sin(X[:, 0] * 2) + X[:, 1] * X[:, 2] + 0.3 * rng.normal(size=n)).astype(np.float64)
    rf = RandomForestRegressor(n_estimators=n_estimators, max_depth=max_depth, max_features=0.5, random_state=seed, n_jobs=-1).fit(X, y)
    return rf, X


def test_importer_sparse_stream_equals_padded_stream_and_sklearn():
    rf, X = _rf(12, 9)
    sp = ddt.importer.from_sklearn(rf, sparse=True)
    pf = ddt.importer.from_sklearn(rf)
    assert sp.sparse and not pf.sparse and sp.num_levels == pf.num_levels
    xs = O.tuples_from_float(X[:1500])
    s = O.SparseModel(O.make_sparse_params(sp.num_trees, sp.num_levels, sp.num_features, cmp_mode=1), sp.node_lines, sp.tree_first_line)
    assert s.check() == 0
    m = O.Model(O.make_params(pf.num_trees, pf.num_levels, pf.num_features, cmp_mode=1), pf.wlines, pf.flines)
    a, gold = O.score_sparse(s, xs, want_gold=True)
    assert np.array_equal(_bits(a), _bits(O.score(m, xs)))
    assert np.array_equal(_bits(O.score(O.sparse_to_perfect(s), xs)), _bits(a))
    ref = rf.predict(X[:1500])
    assert np.max(np.abs(gold - ref)) <= 1e-5 * max(1.0, np.max(np.abs(ref)))
    # a sparse stream is far smaller than the padded one for ragged trees
    assert sp.node_lines.nbytes < pf.wlines.nbytes + pf.flines.nbytes


def test_importer_flushes_subnormal_and_negative_zero_leaves():
    assert ddt.importer.leaf_f32(-1e-50).view(np.uint32) == 0
    assert ddt.importer.leaf_f32(-0.0).view(np.uint32) == 0
    assert ddt.importer.leaf_f32(1e-40).view(np.uint32) == 0
    assert ddt.importer.leaf_f32(-2.5) == np.float32(-2.5)


# ------------------------------------------------------------------------------------------------ GPU
@pytest.fixture(scope="module")
def eng():
    e = ddt.Engine(0)
    yield e
    e.close()


def _gpu_sparse(eng, s, x, sum_mode=0, shard=(0, 1), top=-1, order=0):
    import torch

    q = s.params
    eng.set_option("sparse_top_levels", top)
    eng.set_option("sparse_deep_order", order)
    p = ddt.make_sparse_params(q.num_trees, q.num_levels, q.num_features, q.missing_bits, q.cmp_mode, q.clusters_per_tuple, sum_mode)
    eng.load_model_sparse(p, s.node_lines, s.first, *shard)
    d = torch.from_numpy(np.ascontiguousarray(x).view(np.int32)).cuda()
    out = eng.score_device(d)
    torch.cuda.synchronize()
    return out.cpu().numpy()


@pytest.mark.gpu
@pytest.mark.parametrize("shape", SHAPES)
def test_gpu_sparse_bit_exact_all_top_levels_and_orders(eng, shape):
    T, D, F, full, pm, dist, rows = shape
    for cmp_mode in (0, 1):
        s = O.gen_sparse_model(T, D, F, full, pm, dist, cmp_mode=cmp_mode)
        x = O.gen_tuples(0, rows, F, dist)
        want = O.score_sparse(s, x)
        for top in (-1, 6, 7, 8, 9, 10):
            for order in (0, 1):
                # rank-quantised kernels (u16 tile of the q16 pre-pass), the fp32-tile kernels with dense level K (all top levels as 8-byte
                # records in LDS) and with 16-byte level K-1 records
                for ranked, dense in ((1, 1), (0, 1), (0, 0)):
                    eng.set_option("sparse_q16", ranked)
                    eng.set_option("sparse_dk", dense)
                    try:
                        got = _gpu_sparse(eng, s, x, top=top, order=order)
                    except ddt.DDTError as ex:  # a forced K whose top images do not fit the LDS next to the feature tile
                        assert ex.code == -5 and top == 10 and F > 60
                        continue
                    name = eng.info().variant_name.decode()
                    # (dense level K includes its forms with dense mid levels / dense pair records, chosen by the forest's fill)
                    assert ranked or not name.startswith(("sparse_q_", "sparse_qd_", "sparse_qp_")), (name, ranked, top)
                    assert dense or not name.startswith(("sparse_dk_", "sparse_qd_", "sparse_dm", "sparse_dp_", "sparse_qp_")), (name, dense, top)
                    assert not (ranked and top == -1 and F <= 64) or name.startswith(("sparse_q_", "sparse_qd_", "sparse_qp_")), (name, ranked, top)
                    assert not (not ranked and dense and top in (-1, 6, 7, 8, 9) and F <= 32) or name.startswith(("sparse_dk_", "sparse_dm", "sparse_dp_")), (name, top)
                    assert np.array_equal(_bits(got), _bits(want)), (shape, cmp_mode, top, order, name)
        eng.set_option("sparse_dk", 1)
        eng.set_option("sparse_q16", 1)
        want64 = O.score_sparse(s, x, sum_mode=O.SUM_F64_SEQ)
        assert np.array_equal(_bits(_gpu_sparse(eng, s, x, sum_mode=1)), _bits(want64))
        assert np.array_equal(_bits(eng.score(x)), _bits(want64))  # host feeder path on the loaded model
    info = eng.info()
    assert info.variant_name.decode().startswith("sparse_") and info.local_trees == T
    assert info.model_bytes_unpadded == s.n_lines * 16


@pytest.mark.gpu
def test_gpu_every_sparse_kernel_variant(eng):
    """every compiled (K, trees in lock-step, tile) geometry of the sparse kernel, forced by id, against the oracle"""
    import torch

    for (T, D, F, full, pm, dist, rows) in [(20, 13, 24, 4, 650, 1, 1100), (12, 16, 64, 7, 700, 0, 700)]:
        s = O.gen_sparse_model(T, D, F, full, pm, dist)
        x = O.gen_tuples(0, rows, F, dist)
        want = O.score_sparse(s, x)
        d = torch.from_numpy(x.view(np.int32)).cuda()
        ran = 0
        eng.set_option("variant", -1)
        eng.load_model_sparse(ddt.make_sparse_params(T, D, F), s.node_lines, s.first)   # (forcing a variant re-packs the LOADED model: this one)
        for vid, name in enumerate(ddt.variant_names()):
            if not name.startswith("sparse_"):
                continue
            try:
                eng.set_option("variant", vid)
                eng.load_model_sparse(ddt.make_sparse_params(T, D, F), s.node_lines, s.first)
            except ddt.DDTError as ex:
                assert ex.code == -5, name  # its top images do not fit the LDS next to this feature tile
                continue
            assert eng.info().variant_name.decode() == name
            got = eng.score_device(d)
            torch.cuda.synchronize()
            assert np.array_equal(_bits(got.cpu().numpy()), _bits(want)), (name, T, D, F)
            ran += 1
        eng.set_option("variant", -1)
        assert ran >= 10


@pytest.mark.gpu
@pytest.mark.parametrize("F", [600, 2048])
def test_gpu_sparse_tuples_too_wide_for_a_feature_tile(eng, F):
    """More than ~540 tuple words: no feature tile fits next to the top images, the engine takes `sparse_gf_k6_u8_t256` (features
    gathered from the tuple's row in global memory) -- both comparators, missing values with both directions, a ragged last block,
    all three sums, the host feeder."""
    T, D, rows = 19, 12, 777
    for cmp_mode in (0, 1):
        s = O.gen_sparse_model(T, D, F, 3, 650, 1, cmp_mode=cmp_mode)
        x = O.gen_tuples(5, rows, F, 1)
        used = np.unique(s.node_lines[:, 1] & 0x7FF)
        x[::5, used[0]] = s.params.missing_bits
        x[1::9, used[-1]] = s.params.missing_bits
        for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (1, O.SUM_F64_SEQ), (2, O.SUM_REF_FLOPOCO)):
            got = _gpu_sparse(eng, s, x, sum_mode=sum_mode)
            assert eng.info().variant_name.decode() == "sparse_gf_k6_u8_t256"
            want = O.score_sparse(s, x, sum_mode=ref)
            assert np.array_equal(_bits(got), _bits(want)), (F, cmp_mode, sum_mode)
        assert np.array_equal(_bits(eng.score(x)), _bits(want))


@pytest.mark.gpu
def test_gpu_sparse_tree_shards_and_chain(eng):
    import torch

    s = O.gen_sparse_model(21, 13, 24, 5, 650, 1)
    x = O.gen_tuples(3, 1500, 24, 1)
    for G in (2, 3, 8, 12):  # 12 shards of ceil(21/12) = 2 trees: the last one is EMPTY and scores +0
        parts = np.stack([_gpu_sparse(eng, s, x, shard=(g, G)) for g in range(G)])
        per = (21 + G - 1) // G
        assert not parts[G - 1].any() if (G - 1) * per >= 21 else True
        got = eng.chain_sum_device(torch.from_numpy(parts).cuda()).cpu().numpy()
        assert np.array_equal(_bits(got), _bits(O.score_sparse(s, x, n_devices=G))), G


@pytest.mark.gpu
def test_gpu_sparse_single_leaf_trees_and_clusters(eng):
    # tree 1 is a single leaf (one line, both flags set); cluster counts change the summation order
    lines = np.array([[np.float32(0.5).view(np.uint32), 0x0003 | 0x8000, 1, np.float32(0.25).view(np.uint32)],
                      [np.float32(0.1).view(np.uint32), 0x0001 | 0xC000, np.float32(-1.5).view(np.uint32), np.float32(2.0).view(np.uint32)],
                      [0, 0xC000, np.float32(7.0).view(np.uint32), np.float32(7.0).view(np.uint32)]], np.uint32)
    first = np.array([0, 2, 3], np.uint64)
    x = O.gen_tuples(0, 300, 4, 0)
    for C_ in (1, 2, 4, 8):
        s = O.SparseModel(O.make_sparse_params(2, 2, 4, clusters=C_), lines, first)
        assert s.check() == 0
        assert np.array_equal(_bits(_gpu_sparse(eng, s, x)), _bits(O.score_sparse(s, x)))


@pytest.mark.gpu
def test_gpu_sparse_classes(eng):
    """one-vs-all classes in a sparse stream: per-class sums bit-exact, labels exact; class shards + chain add; host path"""
    import torch

    for (T, D, F, K, inter) in [(30, 10, 12, 3, True), (24, 7, 9, 4, False), (70, 13, 33, 10, True)]:
        s = O.gen_sparse_model(T, D, F, 3, 650, 1, clusters=1)
        x = O.gen_tuples(5, 1500, F, 1)
        want_l, want_s = O.classify_sparse(s, x, K, inter)
        p = ddt.make_sparse_params(T, D, F, clusters=1)
        eng.load_model_sparse(p, s.node_lines, s.first, 0, 1, K, inter)
        d = torch.from_numpy(x.view(np.int32)).cuda()
        gl, gs = eng.classify_device(d)
        torch.cuda.synchronize()
        assert np.array_equal(gl.cpu().numpy(), want_l) and np.array_equal(_bits(gs.cpu().numpy()), _bits(want_s)), (T, K, inter)
        hl, hs = eng.classify(x, want_scores=True)
        assert np.array_equal(hl, want_l) and np.array_equal(_bits(hs), _bits(want_s))
        with pytest.raises(ddt.DDTError):
            eng.score_device(d)  # the scalar call refuses a model with classes
        # two class shards, chain-added per class, then the argmax
        parts = []
        for g in range(2):
            eng.load_model_sparse(p, s.node_lines, s.first, g, 2, K, inter)
            parts.append(eng.classify_device(d, want_labels=False)[1])
        comb = torch.stack([eng.chain_sum_device(torch.stack([parts[0][k], parts[1][k]])) for k in range(K)])
        lab = eng.argmax_device(comb.contiguous())
        wl2, ws2 = O.classify_sparse(s, x, K, inter, n_devices=2)
        assert np.array_equal(lab.cpu().numpy(), wl2) and np.array_equal(_bits(comb.cpu().numpy()), _bits(ws2))


@pytest.mark.gpu
def test_gpu_real_random_forest_classifier_sparse(eng):
    """a scikit-learn RandomForestClassifier (deep, ragged trees) imported as a sparse one-vs-all model: class scores bit-exact
    with the oracle, labels equal to scikit-learn's predict wherever its winning margin is not a rounding tie"""
    import torch
    from sklearn.ensemble import RandomForestClassifier

    rng = np.random.default_rng(3)
    X = rng.normal(size=(6000, 20)).astype(np.float32)
    y = (np.digitize(X[:, 0] + 0.5 * X[:, 1] * X[:, 2], [-0.8, 0.0, 0.8])).astype(np.int64)  # 4 classes
    rf = RandomForestClassifier(n_estimators=40, max_depth=14, random_state=1, n_jobs=-1).fit(X, y)
    im = ddt.importer.from_sklearn(rf, sparse=True)
    assert im.sparse and im.num_classes == 4 and im.num_trees == 160
    Xt = rng.normal(size=(3000, 20)).astype(np.float32)
    xs = O.tuples_from_float(Xt)
    im.load_into(eng)
    assert eng.num_classes == 4
    gl, gs = eng.classify_device(torch.from_numpy(xs.view(np.int32)).cuda())
    torch.cuda.synchronize()
    s = O.SparseModel(O.make_sparse_params(160, im.num_levels, 20, cmp_mode=1, clusters=1), im.node_lines, im.tree_first_line)
    want_l, want_s = O.classify_sparse(s, xs, 4)
    assert np.array_equal(gl.cpu().numpy(), want_l) and np.array_equal(_bits(gs.cpu().numpy()), _bits(want_s))
    proba = rf.predict_proba(Xt)
    assert np.max(np.abs(gs.cpu().numpy().T.astype(np.float64) - proba)) <= 1e-5
    top2 = np.sort(proba, axis=1)
    clear = top2[:, -1] - top2[:, -2] > 1e-5
    assert clear.mean() > 0.9 and np.array_equal(gl.cpu().numpy()[clear], rf.predict(Xt)[clear])


@pytest.mark.gpu
def test_gpu_sparse_rejects_what_the_format_forbids(eng):
    s = O.gen_sparse_model(4, 8, 10, 3, 600, 0)
    p = ddt.make_sparse_params(4, 8, 10)
    k = int(np.nonzero((s.node_lines[:, 1] & 0x4000) == 0)[0][0])
    for mutate, code in [(lambda a: a.__setitem__((k, 2), 0), -1),            # child not after its parent
                         (lambda a: a.__setitem__((0, 1), a[0, 1] | 0x7FF), -1),  # feature index >= F
                         (lambda a: a.__setitem__((0, 1), a[0, 1] | 0x10000), -1)]:  # word 1 [31:16] != 0
        bad = s.node_lines.copy()
        mutate(bad)
        with pytest.raises(ddt.DDTError) as ex:
            eng.load_model_sparse(p, bad, s.first)
        assert ex.value.code == code
    with pytest.raises(ddt.DDTError) as ex:  # deeper than the announced bound
        eng.load_model_sparse(ddt.make_sparse_params(4, 3, 10), s.node_lines, s.first)
    assert ex.value.code == -1
    # leaves outside the exact domain of the reference adder: refused in the reference-order sum, accepted on request
    bad = s.node_lines.copy()
    j = int(np.nonzero(bad[:, 1] & 0x4000)[0][0])
    bad[j, 2] = 0x80000000  # -0
    with pytest.raises(ddt.DDTError) as ex:
        eng.load_model_sparse(p, bad, s.first)
    assert ex.value.code == -5
    eng.set_option("leaf_domain_check", 0)
    eng.load_model_sparse(p, bad, s.first)
    eng.set_option("leaf_domain_check", 1)
    eng.load_model_sparse(ddt.make_sparse_params(4, 8, 10, sum_mode=1), bad, s.first)  # fp64 accumulate: no claim, accepted


@pytest.mark.gpu
def test_gpu_real_random_forest_512_depth16_64_features(eng):
    """BASELINE config 4 as a REAL forest: RandomForestRegressor(512 trees, max_depth 16) on 64 features, imported
    without padding, scored bit-exact against the sparse oracle and within 1e-5 of scikit-learn's own prediction."""
    import torch

    rf, X = _rf(512, 16, F=64, n=8000, seed=4)
    im = ddt.importer.from_sklearn(rf, sparse=True)
    assert im.num_levels == 16 and im.num_trees == 512
    n_nodes = im.node_lines.shape[0]
    assert n_nodes < 512 * 8000  # nowhere near the 2^16 internal nodes per tree of the padded form
    rng = np.random.default_rng(9)
    Xt = rng.normal(size=(6000, 64)).astype(np.float32)
    Xt[rng.random(Xt.shape) < 0.01] = np.nan  # missing values (canonical quiet NaN) exercise the per-node default direction
    xs = O.tuples_from_float(Xt)
    assert (xs == 0x7FC00000).any()
    s = O.SparseModel(O.make_sparse_params(512, 16, 64, cmp_mode=1), im.node_lines, im.tree_first_line)
    want, gold = O.score_sparse(s, xs, want_gold=True)  # bit-level FloPoCo adder model
    im.load_into(eng)
    got = eng.score_device(torch.from_numpy(xs.view(np.int32)).cuda()).cpu().numpy()
    assert np.array_equal(_bits(got), _bits(want)), eng.info().variant_name
    clean = ~np.isnan(Xt).any(axis=1)
    ref = rf.predict(Xt[clean])
    assert np.max(np.abs(gold[clean] - ref)) <= 1e-5 * max(1.0, np.max(np.abs(ref)))
    assert np.max(np.abs(got[clean].astype(np.float64) - ref)) <= 1e-5 * max(1.0, np.max(np.abs(ref)))
