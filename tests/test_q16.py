"""The rank-quantised path (q16 kernels): exactness on adversarial values, fallback rules, timing counters.

q16 replaces every feature by its rank among the model's thresholds for that feature, so values that sit exactly
on, just below and just above thresholds are the interesting inputs; the generic parity suite
(test_gpu_parity.py) already runs every q16 variant on every shape it accepts."""
import numpy as np
import pytest

from oracle import oracle as O
import ddt

pytestmark = pytest.mark.gpu


def _prepass(e, groups):
    """groups = -1: transpose + rank kernels; 0: automatic; 1 / 2 / 4 / 8: that many feature groups (takes effect at the next load)."""
    e.set_option("q16_fused_prepass", 0 if groups < 0 else 1)
    e.set_option("q16_grouped_prepass", 0 if groups < 0 else 1)
    e.set_option("q16_prepass_groups", max(groups, 0))


def _variant(name):
    return ddt.variant_names().index(name)


def _params(m, sum_mode=0):
    p = m.params
    return ddt.make_params(p.num_trees, p.num_levels, p.num_features, p.missing_bits, p.cmp_mode, p.clusters_per_tuple, sum_mode)


# cluster-major image + one accumulator / levels 0-1 from SGPRs / leaves gathered from global memory / leaves staged in LDS
@pytest.mark.parametrize("kernel", ["q16_d8_c8_u4_gl_s2_cm_x", "q16_d8_c8_u4_gl_s2", "q16_d8_c8_u4_gl"])
@pytest.mark.parametrize("cmp_mode", [0, 1])
def test_values_on_and_next_to_thresholds(cmp_mode, kernel):
    T, D, F, n = 200, 8, 32, 4096
    m = O.gen_model(T, D, F, dist=1, cmp_mode=cmp_mode)
    rng = np.random.default_rng(0)
    thr = m.wlines.reshape(T, -1)[:, :255].reshape(-1)             # every threshold bit pattern of the model
    x = O.gen_tuples(0, n, F, dist=1)
    pick = thr[rng.integers(0, thr.size, (n, F))]
    delta = rng.integers(-1, 2, (n, F)).astype(np.int64)            # exactly on / one ulp either side
    near = (pick.astype(np.int64) + delta).astype(np.uint32)
    mask = rng.random((n, F)) < 0.6
    x[:, :F] = np.where(mask, near, x[:, :F])
    x[0, :F] = [0x80000000, 0x00000000, 0x7F800000, 0xFF800000, 0x7FC00001, 0x7FFFFFFF, 0x80000001, 0x00000001] * 4
    e = ddt.Engine(0)
    e.set_option("variant", _variant(kernel))
    want = O.score(m, x)
    for sum_mode in (0, 1, 2):
        if "_cm" in kernel and sum_mode == 1:  # the fp64 sum runs in stream order: a cluster-major image is refused for it
            with pytest.raises(ddt.DDTError):
                e.load_model(_params(m, sum_mode), m.wlines, m.flines)
            continue
        e.load_model(_params(m, sum_mode), m.wlines, m.flines)
        assert e.info().variant_name.decode() == kernel
        got = e.score(x)
        ref = O.score(m, x, sum_mode=(O.SUM_REF_NATIVE, O.SUM_F64_SEQ, O.SUM_REF_FLOPOCO)[sum_mode])
        assert np.array_equal(got.view(np.uint32), ref.view(np.uint32)), (kernel, sum_mode)
    e.close()


@pytest.mark.parametrize("T,clusters", [(1000, 8), (999, 8), (520, 4), (130, 2), (129, 8), (17, 8), (8, 4), (3, 8), (250, 1)])
def test_cluster_major_image_equals_the_ring(T, clusters):
    """`_cm` kernels: the PU groups of a cluster are stored together and one accumulator + a running total replace the ring of C
    accumulators; every (tree count, cluster count) -- partial last group, fewer groups than clusters, one cluster -- must give the
    reference-order sums bit for bit, with IEEE adds and with the reference adder, also on tiles with missing values."""
    D, F, n = 8, 32, 1200
    m = O.gen_model(T, D, F, dist=1, clusters=clusters)
    x = O.gen_tuples(1, n, F, dist=1)
    e = ddt.Engine(0)
    e.set_option("variant", _variant("q16_d8_c8_u4_gl_s2_cm_x"))
    for sum_mode, ref_mode in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        e.load_model(_params(m, sum_mode), m.wlines, m.flines)
        assert e.info().variant_name.decode() == "q16_d8_c8_u4_gl_s2_cm_x"
        assert np.array_equal(e.score(x).view(np.uint32), O.score(m, x, sum_mode=ref_mode).view(np.uint32)), (T, clusters, sum_mode)
    e.close()


def test_auto_selection_and_fallbacks():
    e = ddt.Engine(0)
    w, f = ddt.synth_model(1000, 8, 32)
    e.load_model(ddt.make_params(1000, 8, 32), w, f)
    assert e.info().variant_name.decode() == "q16_d8_c8_u4_gl_s2_cm_x"   # many trees: the pre-pass pays off (cluster-major image, pinned read order)
    e.load_model(ddt.make_params(1000, 8, 32), w, f, 3, 8)          # 125 trees per engine (8-way shard): all rank tables
    assert e.info().variant_name.decode() == "q16_d8_c8_u4_gl_s2_cm_x"   # fit LDS together -> fused pre-pass -> q16 still pays
    _prepass(e, -1)                                                 # with the transpose + rank kernels the fixed pre-pass cost is too high
    e.load_model(ddt.make_params(1000, 8, 32), w, f, 3, 8)
    assert e.info().variant_name.decode() == "d8_t1024_r1_c4_u4_dma_f"
    _prepass(e, 0)
    e.load_model(ddt.make_params(1000, 8, 32), w, f, 0, 12)         # 84 trees x 8 levels >= 480: q16 with the LDS-resident pre-pass
    assert e.info().variant_name.decode() == "q16_d8_c8_u4_gl_s2_cm_x" and e.info().prepass_groups in (1, 2, 4, 8)
    e.load_model(ddt.make_params(1000, 8, 32), w, f, 0, 20)         # 50 trees: below the break-even either way
    assert e.info().prepass_groups == 0
    with pytest.raises(ddt.DDTError):
        e.set_option("q16_prepass_groups", 3)                       # 0, 1, 2, 4 or 8
    assert e.info().variant_name.decode() == "d8_t1024_r1_c4_u4_dma_f"
    # too many distinct thresholds on one feature for 16-bit ranks: 2000 trees x 255 nodes on 4 features (the kernels that cannot score in
    # parts refuse the model)
    w, f = ddt.synth_model(2000, 8, 4)
    e.set_option("variant", _variant("q16_d8_c8_u4_gl"))
    with pytest.raises(ddt.DDTError) as ei:
        e.load_model(ddt.make_params(2000, 8, 4), w, f)
    assert ei.value.code == -5
    e.set_option("variant", -1)
    e.load_model(ddt.make_params(2000, 8, 4), w, f)
    assert e.info().variant_name.decode() == "q16_d8_c8_u4_gl_s2_cm_x"   # round 4: the cluster-major kernel scores such an ensemble in parts
    x = O.gen_tuples(0, 1500, 4)
    m = O.Model(O.make_params(2000, 8, 4), w, f)
    assert np.array_equal(e.score(x).view(np.uint32), O.score(m, x).view(np.uint32))
    e.close()


def test_kernel_timing_counters():
    import torch

    e = ddt.Engine(0)
    w, f = ddt.synth_model(1000, 8, 32)
    e.load_model(ddt.make_params(1000, 8, 32), w, f)
    d = e.synth_tuples_device(0, 1 << 20, 32)
    e.set_option("kernel_timing", 1)
    e.score_device(d)
    st = e.stats()
    assert st.last_score_ms > 0.2 and 0.0 < st.last_prepass_ms < st.last_score_ms
    e.set_option("variant", _variant("d8_t1024_r1_c4_u4_dma_f"))
    e.score_device(d)
    st = e.stats()
    assert st.last_score_ms > 0.2 and st.last_prepass_ms < 0.05    # fp32 kernel: no pre-pass
    # a loop of calls without a host synchronisation in between: the library keeps the events of up to 64 launches and folds them
    # into the sums when the counters are read; more than 64 in flight only makes a launch wait for the oldest
    e.set_option("variant", -1)
    n0, s0, p0 = st.timed_launches, st.sum_score_ms, st.sum_prepass_ms
    assert n0 == 2
    for _ in range(70):
        e.score_device(d)
    st = e.stats()
    assert st.timed_launches == n0 + 70
    per_launch = (st.sum_score_ms - s0) / 70
    assert 0.5 * st.last_score_ms < per_launch < 2.0 * st.last_score_ms and st.sum_prepass_ms > p0
    e.set_option("kernel_timing", 0)
    e.score_device(d)
    assert e.stats().timed_launches == n0 + 70
    torch.cuda.synchronize()
    e.close()


@pytest.mark.parametrize("cmp_mode", [0, 1])
@pytest.mark.parametrize("shape", ["cluster+outliers", "two_keys", "one_key", "full_range"])
def test_rank_search_on_degenerate_threshold_distributions(cmp_mode, shape):
    """The rank pre-pass looks a value up in equal slices of the feature's key range, then finishes with a short
    binary search; clustered thresholds with far outliers put almost every key in one slice (the search then
    degenerates to the plain one), one or two distinct keys exercise the zero-width and tiny ranges, and
    full_range spans negative to positive bit patterns (uint32 difference > 2^31)."""
    T, D, F, n = 64, 8, 8, 5000
    m = O.gen_model(T, D, F, dist=1, cmp_mode=cmp_mode)
    rng = np.random.default_rng(7)
    w = m.wlines.copy().reshape(T, -1)
    nint = 255
    if shape == "cluster+outliers":
        thr = (np.float32(0.25) + rng.integers(0, 3000, (T, nint)).astype(np.float32) * np.float32(2 ** -22)).astype(np.float32)
        thr[0, 0], thr[1, 0], thr[2, 0] = np.float32(-3.0e38), np.float32(3.0e38), np.float32(1e-30)
    elif shape == "two_keys":
        thr = np.where(rng.random((T, nint)) < 0.5, np.float32(0.5), np.float32(-0.5)).astype(np.float32)
    elif shape == "one_key":
        thr = np.full((T, nint), np.float32(0.125), np.float32)
    else:
        thr = rng.integers(0, 2 ** 32, (T, nint), dtype=np.uint64).astype(np.uint32).view(np.float32)
        thr = np.where(np.isnan(thr), np.float32(1.0), thr).astype(np.float32)
    w[:, :nint] = thr.view(np.uint32)
    m = O.Model(m.params, w.reshape(m.wlines.shape), m.flines)
    x = O.gen_tuples(5, n, F, dist=1, missing_bits=m.params.missing_bits)
    flat = thr.view(np.uint32).reshape(-1)
    pick = flat[rng.integers(0, flat.size, (n, F))].astype(np.int64) + rng.integers(-2, 3, (n, F))
    mask = rng.random((n, F)) < 0.7
    x[:, :F] = np.where(mask, (pick & 0xFFFFFFFF).astype(np.uint32), x[:, :F])
    e = ddt.Engine(0)
    e.set_option("variant", _variant("q16_d8_c8_u4_gl"))
    want = O.score(m, x)
    for groups in (0, 2, 8, -1):  # the LDS-resident pre-pass (segmented bucket index) in 1 / 2 / 8 feature groups; transpose + rank kernels
        _prepass(e, groups)
        e.load_model(_params(m), m.wlines, m.flines)
        assert e.info().variant_name.decode() == "q16_d8_c8_u4_gl"
        got = e.score(x)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), groups
    e.close()


@pytest.mark.parametrize("T", [100, 200, 300])   # fused pre-pass, fused in two feature groups, transpose + rank kernels
@pytest.mark.parametrize("n", [1, 3, 700, 1025, 1500, 2049])
def test_short_batches_do_not_read_past_the_tuples(n, T):
    """Batches that end well inside a 1024-tuple tile: the pre-pass pads q to whole tiles but must not read the
    tuple buffer past row n (the host path hands it an exactly sized device allocation)."""
    D, F = 8, 32
    m = O.gen_model(T, D, F, dist=1)
    x = O.gen_tuples(21, n, F, dist=1)
    e = ddt.Engine(0)
    e.set_option("variant", _variant("q16_d8_c8_u4_gl"))
    e.load_model(_params(m), m.wlines, m.flines)
    e.set_option("feeder_rows", 1 << 20)
    got = e.score(x)
    assert np.array_equal(got.view(np.uint32), O.score(m, x).view(np.uint32))
    e.close()


@pytest.mark.parametrize("cmp_mode", [0, 1])
def test_fused_and_two_kernel_prepass_agree(cmp_mode):
    """Small tables: the fused pre-pass (all tables in LDS) and the transpose + rank kernels must give identical
    scores, missing values and negatives included; ragged batch (last tile partly filled)."""
    T, D, F, n = 100, 8, 28, 5 * 1024 + 333
    m = O.gen_model(T, D, F, dist=1, cmp_mode=cmp_mode)
    x = O.gen_tuples(31, n, F, dist=1)
    want = O.score(m, x)
    e = ddt.Engine(0)
    e.set_option("variant", _variant("q16_d8_c8_u4_gl"))
    for groups in (1, 2, 4, 8, -1):
        _prepass(e, groups)
        e.load_model(_params(m), m.wlines, m.flines)
        got = e.score(x)
        assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"groups={groups}"
    e.close()


@pytest.mark.parametrize("T,F", [(1000, 32), (400, 32), (700, 20), (900, 12), (320, 7)])
def test_grouped_and_two_kernel_prepass_agree(T, F):
    """Big tables (they do not fit one CU's LDS together): the grouped pre-pass (one launch, blocks split over feature
    groups x row partitions, no transposed fp32 intermediate) and the transpose + rank kernels must give identical
    scores -- missing values and negatives included, narrow tuples (short last group), batches of fewer than 8 tiles (one
    row partition) and of more (8 partitions), ragged tails."""
    D = 8
    m = O.gen_model(T, D, F, dist=1)
    e = ddt.Engine(0)
    e.set_option("variant", _variant("q16_d8_c8_u4_gl"))
    for n in (1, 1500, 11 * 1024 + 77):
        x = O.gen_tuples(41 + n, n, F, dist=1, missing_bits=m.params.missing_bits)
        want = O.score(m, x)
        for groups in (0, 4, 8, -1):
            _prepass(e, groups)
            e.load_model(_params(m), m.wlines, m.flines)
            got = e.score(x)
            assert np.array_equal(got.view(np.uint32), want.view(np.uint32)), f"groups={groups} n={n}"
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("T,F,clusters,n", [(650, 4, 1, 5000), (1100, 4, 8, 3000), (4000, 16, 8, 2500)])
def test_ensembles_beyond_the_u16_rank_range_are_scored_in_parts(T, F, clusters, n):
    """More than 38848 distinct thresholds on a feature (4000 trees x 255 nodes over 16 features: ~64 k each): the cluster-major kernel
    scores the ensemble in parts with rank tables of their own, the reference-order sum handed from launch to launch -- bit-exact, both adders."""
    import torch

    D = 8
    m = O.gen_model(T, D, F, dist=0, clusters=clusters)
    x = O.gen_tuples(7, n, F, dist=0)
    x[11, 2] = 0x7FC00000                                                    # a tile with a missing value: the parts' slow images
    e = ddt.Engine(0)
    d = torch.from_numpy(x.view(np.int32)).cuda()
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        e.load_model(_params(m, sum_mode), m.wlines, m.flines)
        assert e.info().variant_name.decode() == "q16_d8_c8_u4_gl_s2_cm_x"
        before = e.stats().kernel_launches
        got = e.score_device(d)
        torch.cuda.synchronize()
        assert e.stats().kernel_launches - before >= 2                       # at least two parts
        want = O.score_fast(m, x, sum_mode=ref)
        assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), (T, clusters, sum_mode)
        assert np.array_equal(e.score(x).view(np.uint32), want.view(np.uint32))   # host feeder path
    e.close()


@pytest.mark.gpu
@pytest.mark.parametrize("K,parts", [(32767, 1), (32768, 1), (35001, 1), (37727, 1), (38848, 1), (38849, 2)])
def test_rank_tables_of_up_to_38848_keys_fit_one_blocks_lds(K, parts):
    """One rank table = what ONE block of rank_kernel holds in LDS (csrc/ddt_engine_priv.h kQ16MaxTable: 38848 keys; 32767 and a
    power-of-two table until round 6).  A feature with EXACTLY K distinct thresholds: the table's padding (a multiple of 32 entries beyond
    32767 keys), the probes' clamp at its last entry, ranks above 32767 in the u16 tile and in the node records; values on every key, one
    code below and above, beyond both ends, missing.  One part up to 38848 keys, two from 38849; `q16_max_table` = 32767 (the old limit)
    must give the same bits in more parts; the same model as a sparse forest (u16-rank sparse kernels, the same rank kernel)."""
    import torch

    T, D, F, n = 160, 8, 3, 6000
    nint, rng = 255, np.random.default_rng(K)
    keys = np.sort(rng.choice(np.arange(1, 1 << 24, dtype=np.int64), K, replace=False)) * 64 + 0x3D000000   # K distinct positive fp32 patterns
    fidx = np.zeros((T, nint), np.int64)
    thr = np.empty((T, nint), np.uint32)
    flat = np.concatenate([keys, keys[rng.integers(0, K, T * nint - K - 300)]])                               # every key at least once, the rest repeats
    rng.shuffle(flat)
    thr.reshape(-1)[: flat.size] = flat.astype(np.uint32)
    fidx.reshape(-1)[flat.size:] = rng.integers(1, F, 300)                                                    # 300 nodes on the other features
    thr.reshape(-1)[flat.size:] = rng.random(300).astype(np.float32).view(np.uint32)
    leaves = ((rng.integers(1, 1 << 20, (T, 1 << D)).astype(np.float32) - np.float32(1 << 19)) * np.float32(2.0 ** -24))
    mr = rng.integers(0, 2, (T, nint))
    m = O.pack_model(thr, fidx, mr, leaves, F, clusters=2)
    x = O.gen_tuples(3, n, F, dist=0, missing_bits=m.params.missing_bits)
    col = np.concatenate([keys[rng.integers(0, K, n - 8)] + rng.integers(-1, 2, n - 8), [keys[0] - 1, keys[0], keys[-1], keys[-1] + 1, 0, 0x7F000000, keys[K // 2], keys[32767 % K]]])
    x[:, 0] = col.astype(np.uint32)
    x[17, 0] = m.params.missing_bits
    d = torch.from_numpy(x.view(np.int32)).cuda()
    e = ddt.Engine(0)
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        want = O.score_fast(m, x, sum_mode=ref)
        for limit in (38848, 32767):
            e.set_option("q16_max_table", limit)
            e.set_option("q16_cluster_split", 0)                                # (count the parts, not the slices of a small batch)
            e.load_model(_params(m, sum_mode), m.wlines, m.flines)
            assert e.info().variant_name.decode() == "q16_d8_c8_u4_gl_s2_cm_x"
            before = e.stats().kernel_launches
            got = e.score_device(d)
            torch.cuda.synchronize()
            launches = e.stats().kernel_launches - before
            assert launches == (parts if limit == 38848 else 1 if K <= 32767 else 2), (K, limit, launches)
            assert np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)), (K, sum_mode, limit)
            assert np.array_equal(e.score(x).view(np.uint32), want.view(np.uint32))
    with pytest.raises(ddt.DDTError):
        e.set_option("q16_max_table", 38849)
    e.set_option("q16_max_table", 38848)
    s = O.sparse_from_perfect(m)
    q = s.params
    e.load_model_sparse(ddt.make_sparse_params(q.num_trees, q.num_levels, q.num_features, q.missing_bits, q.cmp_mode, q.clusters_per_tuple, 0), s.node_lines, s.first)
    name = e.info().variant_name.decode()
    assert name.startswith("sparse_q") == (K <= 38848), name                   # u16 ranks while ONE table holds the forest's thresholds (sparse forests have no parts)
    assert np.array_equal(e.score(x).view(np.uint32), O.score_sparse_fast(s, x).view(np.uint32))
    e.close()
