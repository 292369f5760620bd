"""GPU parity tests: the HIP path, called through the C-ABI (include/ddt.h via ddt.Engine), against the
CPU oracle on identical seeded inputs.  Bar: BIT-EXACT fp32 scores (the kernel reproduces the reference's
adder order) for sum_mode 0, bit-exact for the fp64-accumulate mode too, and the
north-star tolerance (1e-6 relative to max(|gold|, sum|leaf|)) against the fp64 gold sum.
"""
import os

import numpy as np
import pytest

from oracle import oracle as O
import ddt
from tests import sharded_ref as SR

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def eng():
    e = ddt.Engine(0)
    yield e
    e.close()


def _params(m, sum_mode=0):
    p = m.params
    return ddt.make_params(p.num_trees, p.num_levels, p.num_features, p.missing_bits, p.cmp_mode,
                           p.clusters_per_tuple, sum_mode)


def _bits(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


def _gpu_score(eng, m, x, sum_mode=0, variant=-1, shard=(0, 1)):
    import torch

    # load with the automatic choice first, then force the kernel: forcing it while the PREVIOUS call's model is still loaded would
    # re-pack that model, and a kernel may refuse its sum mode (the cluster-major "_cm" image and the fp64 sum)
    eng.set_option("variant", -1)
    eng.load_model(_params(m, sum_mode), m.wlines, m.flines, *shard)
    eng.set_option("variant", variant)
    d = torch.from_numpy(x.view(np.int32)).cuda()
    out = eng.score_device(d)
    torch.cuda.synchronize()
    return out.cpu().numpy()


def _fitting_variants(eng, m):
    """ids of every compiled kernel variant that accepts this model (always includes 0 = generic)."""
    ids = []
    for v, _name in enumerate(ddt.variant_names()):
        try:
            eng.set_option("variant", v)
            eng.load_model(_params(m), m.wlines, m.flines)
            ids.append(v)
        except ddt.DDTError as ex:
            assert ex.code == -5  # DDT_EUNSUPPORTED: wrong depth / tile does not fit LDS
    eng.set_option("variant", -1)
    return ids


# BASELINE.json configs (rows reduced to what the oracle scores in seconds), plus edge shapes
SHAPES = [
    # T, D, F, rows, dist
    (8, 4, 16, 1000, 0),      # config 1 exactly
    (100, 6, 28, 6001, 0),    # config 2 shape
    (100, 6, 28, 3000, 1),    #   with negatives + missing values (slow path inside the kernel)
    (1000, 8, 32, 2500, 0),   # config 3 shape (single GPU holds all 1000 trees)
    (1000, 8, 32, 1100, 1),
    (37, 8, 32, 1029, 1),     # tree count not a multiple of 8: EMPTY slots
    (9, 8, 20, 515, 1),       # F not a multiple of 4
    (64, 6, 64, 700, 1),
    (16, 4, 8, 300, 1),
    (90, 7, 20, 3001, 1),     # odd depths have their own tile / stream / rank-quantised variants too
    (70, 5, 33, 2000, 1),
    (300, 5, 16, 1500, 0),
    (200, 3, 12, 2500, 1),
    (600, 7, 32, 1200, 0),
    (120, 8, 64, 1500, 1),    # 64 features: the 512-tuple tile variants
    (90, 8, 200, 700, 1),     # 200 features: 128-tuple tiles
    (90, 6, 150, 700, 1),
    (60, 8, 400, 300, 1),     # 400 features: 64-tuple tiles
    (70, 5, 180, 500, 1),
    (70, 7, 160, 500, 0),
    (40, 4, 200, 400, 1),
    (260, 9, 24, 1500, 1),    # depths 9 and 10: rank-quantised variants with one block per CU
    (230, 10, 32, 1100, 1),
]


@pytest.mark.parametrize("T,D,F,rows,dist", SHAPES)
def test_bit_exact_vs_oracle_all_variants(eng, T, D, F, rows, dist):
    m = O.gen_model(T, D, F, dist=dist)
    x = O.gen_tuples(3, rows, F, dist=dist, missing_bits=m.params.missing_bits)
    want = O.score(m, x, sum_mode=O.SUM_REF_FLOPOCO)
    want64 = O.score(m, x, sum_mode=O.SUM_F64_SEQ)
    _, gold = O.score(m, x, want_gold=True)
    vids = _fitting_variants(eng, m)
    assert 0 in vids and (len(vids) > 1 or D not in (3, 4, 5, 6, 7, 8, 9, 10))
    names = ddt.variant_names()
    want_ieee = O.score(m, x, sum_mode=O.SUM_REF_NATIVE)  # the same order with IEEE adds (differs from `want` only in the expDiff = 25 case)
    for v in vids:
        got = _gpu_score(eng, m, x, 0, v)
        assert np.array_equal(_bits(got), _bits(want_ieee)), f"variant {names[v]} differs from the reference-order sum (IEEE adds)"
        got2 = _gpu_score(eng, m, x, 2, v)
        assert np.array_equal(_bits(got2), _bits(want)), f"variant {names[v]} differs from the reference adder network"
        if "_cm" in names[v]:  # cluster-major image order: refused for the fp64 sum, which is defined on the stream order
            with pytest.raises(ddt.DDTError) as ei:
                _gpu_score(eng, m, x, 1, v)
            assert ei.value.code == -5
            continue
        got64 = _gpu_score(eng, m, x, 1, v)
        assert np.array_equal(_bits(got64), _bits(want64)), f"variant {names[v]} fp64 mode"
    # north-star tolerance against the fp64 gold
    leaves = np.stack([O.leaves(m, x[r]).view(np.float32) for r in range(0, rows, max(1, rows // 50))])
    sabs = np.abs(leaves.astype(np.float64)).sum(axis=1)
    g = gold[:: max(1, rows // 50)]
    assert np.all(np.abs(got[:: max(1, rows // 50)].astype(np.float64) - g) <= 1e-6 * np.maximum(np.abs(g), sabs))


@pytest.mark.parametrize("cmp_mode", [0, 1])
@pytest.mark.parametrize("clusters", [1, 2, 4, 8])
def test_compare_modes_and_cluster_orders(eng, cmp_mode, clusters):
    T, D, F, rows = 53, 8, 32, 777
    m = O.gen_model(T, D, F, dist=1, cmp_mode=cmp_mode, clusters=clusters)
    x = O.gen_tuples(9, rows, F, dist=1)
    x[5, 3] = 0x80000000   # -0.0
    x[6, 4] = 0x7FC00001   # a NaN that is not the missing pattern
    x[7, 5] = 0xFF800000   # -inf
    want = O.score(m, x)
    for v in _fitting_variants(eng, m):
        assert np.array_equal(_bits(_gpu_score(eng, m, x, 0, v)), _bits(want)), (cmp_mode, clusters, v)


@pytest.mark.parametrize("D,F", [(1, 4), (2, 5), (3, 7), (5, 33), (7, 12), (9, 64), (10, 100), (12, 16), (13, 8), (16, 64), (14, 200), (15, 24), (4, 2048), (8, 200)])
def test_generic_kernel_covers_odd_shapes(eng, D, F):
    T = 11 if D < 12 else 3
    m = O.gen_model(T, D, F, dist=1)
    x = O.gen_tuples(1, 401, F, dist=1)
    want = O.score(m, x)
    got = _gpu_score(eng, m, x, variant=0)  # force the generic kernel: depths 3-8 have specialised variants as well
    assert eng.info().variant_name.decode() == "generic"
    assert np.array_equal(_bits(got), _bits(want))
    eng.set_option("variant", -1)


def test_missing_reset_value_and_all_missing(eng):
    # CSR205 reset value 0 => 0.0 is the missing pattern (EngineCSR.sv:167); a tuple of all-missing features
    m = O.gen_model(40, 6, 28, dist=0, missing_bits=0)
    x = O.gen_tuples(0, 600, 28, dist=0)
    x[::7, :28] = 0
    assert np.array_equal(_bits(_gpu_score(eng, m, x)), _bits(O.score(m, x)))


def test_edge_sizes_and_feeder_path(eng):
    m = O.gen_model(100, 6, 28, dist=1)
    eng.set_option("variant", -1)
    eng.load_model(_params(m), m.wlines, m.flines)
    tile = eng.info().tile_tuples
    for n in (1, 3, tile - 1, tile, tile + 1, 4 * tile + 5):
        x = O.gen_tuples(77, n, 28, dist=1)
        assert np.array_equal(_bits(eng.score(x)), _bits(O.score(m, x))), n
    assert eng.score(np.zeros((0, 28), np.uint32)).size == 0
    # pinned double-buffered feeder with many small chunks == one device call
    x = O.gen_tuples(5, 10_000, 28, dist=1)
    eng.set_option("feeder_rows", 777)
    a = eng.score(x)
    eng.set_option("feeder_rows", 1 << 18)
    b = eng.score(x)
    assert np.array_equal(_bits(a), _bits(b)) and np.array_equal(_bits(a), _bits(O.score(m, x)))
    st = eng.stats()
    assert st.tuples_in == st.tuples_out and st.tuples_in >= 20_000 and st.kernel_launches >= 14


def test_error_behaviour(eng):
    w, f = ddt.synth_model(8, 4, 16)
    e2 = ddt.Engine(0)
    with pytest.raises(ddt.DDTError) as ei:
        e2.score(np.zeros((4, 16), np.uint32))  # no model
    assert ei.value.code == -4
    bad = ddt.make_params(8, 4, 16, clusters=3)
    with pytest.raises(ddt.DDTError):
        e2.load_model(bad, w, f)
    with pytest.raises(ddt.DDTError):
        e2.load_model(ddt.make_params(9, 4, 16), w, f)  # stream too short
    f2 = f.copy()
    f2[0] = 17  # feature index >= F
    with pytest.raises(ddt.DDTError):
        e2.load_model(ddt.make_params(8, 4, 16), w, f2)
    e2.close()


def test_golden_fixtures(eng):
    gdir = os.path.join(ROOT, "tests", "golden")
    for fn in sorted(os.listdir(gdir)):
        if not fn.endswith(".npz") or fn.endswith("_rtl_vectors.npz"):
            continue
        g = np.load(os.path.join(gdir, fn))
        T, D, F, miss, wl, fl, cmp_mode, C = [int(v) for v in g["params"]]
        for sum_mode, key in ((0, "score_ref_bits"), (1, "score_f64_bits")):
            p = ddt.make_params(T, D, F, miss, cmp_mode, C, sum_mode)
            eng.set_option("variant", -1)
            eng.load_model(p, g["wlines"], g["flines"])
            assert np.array_equal(_bits(eng.score(g["tuples"])), g[key]), (fn, key)


def test_virtual_ranks_tree_sharded_chain(eng):
    """Tree-sharded mode on one GPU: G engines each hold one shard; chain-add of their partials is
    bit-exact with the oracle's multi-device model and with a real single-engine run to fp32 rounding."""
    import torch

    T, D, F, rows = 1000, 8, 32, 3000
    m = O.gen_model(T, D, F)
    x = O.gen_tuples(0, rows, F)
    d = torch.from_numpy(x.view(np.int32)).cuda()
    for G in (2, 4, 8):
        parts = torch.empty((G, rows), dtype=torch.float32, device="cuda")
        engines = []
        for g in range(G):
            e = ddt.Engine(0)
            e.load_model(_params(m), m.wlines, m.flines, g, G)
            i = e.info()
            assert (i.tree_begin, i.tree_end) == ddt.shard_bounds(T, G)[g]
            e.score_device(d, out=parts[g])
            engines.append(e)
        got = engines[0].chain_sum_device(parts)
        torch.cuda.synchronize()
        assert np.array_equal(_bits(got.cpu().numpy()), _bits(O.score(m, x, n_devices=G))), G
        for g, e in enumerate(engines):
            b, en = ddt.shard_bounds(T, G)[g]
            assert np.array_equal(_bits(parts[g].cpu().numpy()), _bits(O.score_shard(m, x, b, en)))
            e.close()


def test_device_generator_matches_oracle(eng):
    import torch

    for F, dist in ((32, 0), (28, 1), (5, 1)):
        d = eng.synth_tuples_device(1 << 33, 4099, F, dist)
        torch.cuda.synchronize()
        assert np.array_equal(d.cpu().numpy().view(np.uint32), O.gen_tuples(1 << 33, 4099, F, dist=dist))


@pytest.mark.parametrize("T,D,F,N", [(100, 6, 28, 10_000_000), (1000, 8, 32, 4_000_000)])
def test_full_size_properties(eng, T, D, F, N):
    """BASELINE-size batches (config 2 at its full 10 M rows; config 3's per-step batch): properties that do
    not need the oracle to score everything -- determinism, chunk invariance, and exact parity on a
    strided sample of rows that the oracle re-scores."""
    import torch

    w, f = ddt.synth_model(T, D, F)
    p = ddt.make_params(T, D, F)
    eng.set_option("variant", -1)
    eng.load_model(p, w, f)
    d = eng.synth_tuples_device(0, N, F)
    a = eng.score_device(d)
    b = eng.score_device(d)
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))                      # idempotent / deterministic
    half = N // 2 + 13
    c = torch.cat([eng.score_device(d[:half]), eng.score_device(d[half:].contiguous())])
    assert torch.equal(a.view(torch.int32), c.view(torch.int32))                      # batch-split invariant
    idx = torch.arange(0, N, 9973, device="cuda")
    xs = d[idx].cpu().numpy().view(np.uint32)
    m = O.Model(O.make_params(T, D, F), w, f)
    assert np.array_equal(_bits(a[idx].cpu().numpy()), _bits(O.score(m, xs)))        # parity on the sample
    assert np.array_equal(xs, np.concatenate([O.gen_tuples(int(i), 1, F) for i in idx.cpu().numpy()[:50]] +
                                             [xs[50:]]))                              # inputs are the seeded ones
    assert torch.isfinite(a).all()


def test_config3_full_batch_and_eight_way_chain(eng):
    """BASELINE config 3 at its full size (1000 trees x depth 8 x 32 features, 100 M tuples resident in HBM):
    determinism, batch-split invariance, bit-exact parity on a strided sample, and the 8-way tree-sharded job as
    "virtual ranks" -- eight shard engines' partial scores chain-added in the reference's hop order -- against the
    oracle's 8-device model (bit-exact on the sample) and against the single-engine scores (<= 1e-6 relative)."""
    import torch

    T, D, F, N, G = 1000, 8, 32, 100_000_000, 8
    w, f = ddt.synth_model(T, D, F)
    p = ddt.make_params(T, D, F)
    eng.set_option("variant", -1)
    eng.load_model(p, w, f)
    d = eng.synth_tuples_device(0, N, F)
    a = eng.score_device(d)
    b = eng.score_device(d)
    torch.cuda.synchronize()
    assert torch.equal(a.view(torch.int32), b.view(torch.int32))
    del b
    cut = 37_000_001
    c = torch.cat([eng.score_device(d[:cut]), eng.score_device(d[cut:])])
    assert torch.equal(a.view(torch.int32), c.view(torch.int32))
    del c
    idx = torch.arange(0, N, 9973, device="cuda")
    xs = d[idx].cpu().numpy().view(np.uint32)
    m = O.Model(O.make_params(T, D, F), w, f)
    assert np.array_equal(_bits(a[idx].cpu().numpy()), _bits(O.score(m, xs)))
    parts = torch.empty((G, N), dtype=torch.float32, device="cuda")
    for g in range(G):
        eng.load_model(p, w, f, g, G)
        eng.score_device(d, out=parts[g])
    total = eng.chain_sum_device(parts)
    torch.cuda.synchronize()
    assert np.array_equal(_bits(total[idx].cpu().numpy()), _bits(O.score(m, xs, n_devices=G)))
    # ... and on two contiguous 256 K-row stretches (the start and an unaligned interior one) with the fast form of the
    # oracle's 8-device model: per-device reference-order sums (host IEEE adds == the FloPoCo model on these normal
    # values), chain-added host -> dev1 -> ...
    for lo in (0, 61_234_567):
        xs2 = d[lo: lo + 262_144].cpu().numpy().view(np.uint32)
        want = O.score(m, xs2, sum_mode=O.SUM_REF_NATIVE, n_devices=G)
        assert np.array_equal(_bits(total[lo: lo + 262_144].cpu().numpy()), _bits(want)), lo
        assert np.array_equal(_bits(a[lo: lo + 262_144].cpu().numpy()), _bits(O.score_fast(m, xs2))), lo
    # north-star tolerance for regrouped fp32 sums: 1e-6 relative to max(|score|, sum |leaf|); sum |leaf| <= 1000 x 0.1
    assert (total - a).abs().max().item() <= 1e-6 * 100.0
    eng.load_model(p, w, f)


def test_config4_shape_full_forest(eng):
    """BASELINE config 4's shape -- 512 perfect depth-16 trees, 64 features (403 MB of nodes; deep, divergent walks;
    top levels staged in LDS, lower levels gathered from L2/HBM) -- on 1 M rows: determinism and bit-exact parity on a
    strided sample."""
    import torch

    T, D, F, N = 512, 16, 64, 1_000_000
    w, f = ddt.synth_model(T, D, F)
    eng.set_option("variant", -1)
    m = O.Model(O.make_params(T, D, F), w, f)
    d = eng.synth_tuples_device(0, N, F)
    idx = torch.arange(0, N, 997, device="cuda")
    xs = d[idx].cpu().numpy().view(np.uint32)
    want = O.score(m, xs)
    try:
        # round 6: no tuned perfect-tree kernel at depth 16 -> the engine hands the model to the sparse-forest kernels (a perfect tree is a sparse
        # tree whose leaves all sit at depth D; here: 32-bit ranks + pair records); with the switch off it is `generic` as before
        for via_sparse in (1, 0):
            eng.set_option("generic_via_sparse", via_sparse)
            eng.load_model(ddt.make_params(T, D, F), w, f)
            name = eng.info().variant_name.decode()
            assert (name.startswith("sparse_") and not eng.info().fallback_kernel) if via_sparse else name == "generic", name
            a = eng.score_device(d)
            b = eng.score_device(d)
            torch.cuda.synchronize()
            assert torch.equal(a.view(torch.int32), b.view(torch.int32))
            assert np.array_equal(_bits(a[idx].cpu().numpy()), _bits(want)), name
    finally:
        eng.set_option("generic_via_sparse", 1)
    w2, f2 = ddt.synth_model(40, 6, 28)  # leave a small model behind for the tests that follow
    eng.load_model(ddt.make_params(40, 6, 28), w2, f2)


def _two_rank_worker(rank, world, port, mode, ret):
    import torch
    import torch.distributed as dist

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        torch.cuda.set_device(0)
        T, D, F, n = 1000, 8, 32, 5003
        w, f = ddt.synth_model(T, D, F)
        e = ddt.Engine(0)
        e.load_model(ddt.make_params(T, D, F), w, f, rank, world)
        d = e.synth_tuples_device(0, n, F)
        sc = SR.ShardedScorer.from_engine(e, mode=mode, chunk_rows=2048)
        got = sc.score(d)
        torch.cuda.synchronize()
        m = O.Model(O.make_params(T, D, F), w, f)
        want = O.score(m, d.cpu().numpy().view(np.uint32), n_devices=world)
        ret[rank] = bool(np.array_equal(got.cpu().numpy().view(np.uint32), want.view(np.uint32)))
        e.close()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("mode", ["allreduce", "chain"])
def test_two_ranks_tree_sharded_on_one_gpu(mode):
    """One process per rank (here both on cuda:0, gloo as the collective backend because RCCL refuses two ranks
    on one device): the whole N>1 host path -- shard load, chunked partial scoring, combine -- against the
    oracle's 2-device model.  With 2 ranks a sum of two fp32 partials is order independent => bit-exact."""
    import socket
    import torch.multiprocessing as mp

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    ret = ctx.Manager().dict()
    procs = [ctx.Process(target=_two_rank_worker, args=(r, 2, port, mode, ret)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(300)
        assert p.exitcode == 0
    assert ret.get(0) and ret.get(1), dict(ret)


@pytest.mark.parametrize("T,D,F,dist", [(12, 8, 32, 0), (7, 8, 32, 1), (3, 8, 20, 1), (125, 8, 28, 1), (1, 8, 4, 0)])
def test_persistent_tile_kernels_walk_many_tiles(eng, T, D, F, dist):
    """The _p variants keep one block per CU alive and prefetch the next tile's tuples into registers: give
    every block several tiles (rows > 2 x 256 CUs x 1024) plus a ragged tail, with missing values in some tiles
    only when dist=1, and compare EVERY row with the oracle."""
    rows = 2 * 256 * 1024 + 256 * 1024 // 3 + 77
    m = O.gen_model(T, D, F, dist=dist)
    x = O.gen_tuples(11, rows, F, dist=0, missing_bits=m.params.missing_bits)
    if dist:
        x[5000:9000] = O.gen_tuples(12, 4000, F, dist=1, missing_bits=m.params.missing_bits)
        x[-3000:] = O.gen_tuples(13, 3000, F, dist=1, missing_bits=m.params.missing_bits)
    want = O.score(m, x)
    names = ddt.variant_names()
    vids = [v for v in _fitting_variants(eng, m) if names[v].endswith("p")]
    assert vids, "no persistent variant accepted this shape"
    for v in vids:
        got = _gpu_score(eng, m, x, 0, v)
        bad = np.flatnonzero(_bits(got) != _bits(want))
        assert bad.size == 0, f"{names[v]}: {bad.size} rows differ, first {bad[:5]}"
    eng.set_option("variant", -1)


def test_leaves_outside_the_exact_domain_are_refused_unless_asked(eng):
    """The GPU adds are IEEE-754; the reference's FloPoCo adder treats sub-normal / Inf / NaN inputs as normals and keeps
    -0 (FPAdder_2cycles_latency.v:313-320,376-385).  Such leaves are refused in the reference-order sum (the caller flushes
    them, as ddt.importer does), accepted on request and in the fp64-accumulate mode -- where the result is then the
    IEEE one, shown here against the oracle's IEEE-adds mode."""
    import torch

    T, D, F = 16, 4, 8
    m = O.gen_model(T, D, F, 0)
    nint = (1 << D) - 1
    for bad in (0x80000000, 0x00000001, 0x807FFFFF, 0x7F800000, 0x7FC00000):  # -0, sub-normals, +Inf, NaN
        w = m.wlines.copy()
        w[3 * O.wlpt(D) * 4 + nint + 5] = bad  # one leaf of tree 3
        with pytest.raises(ddt.DDTError) as ex:
            eng.load_model(ddt.make_params(T, D, F), w, m.flines)
        assert ex.value.code == -5 and "leaf" in str(ex.value)
        eng.load_model(ddt.make_params(T, D, F, sum_mode=1), w, m.flines)  # fp64 accumulate: no bit-exactness claim
    # on request: IEEE semantics, equal to the oracle's reference-order sum with host IEEE adds
    w = m.wlines.copy()
    w[3 * O.wlpt(D) * 4 + nint + 5] = 0x80000000
    w[5 * O.wlpt(D) * 4 + nint + 2] = 0x00000123
    eng.set_option("leaf_domain_check", 0)
    try:
        eng.load_model(ddt.make_params(T, D, F), w, m.flines)
        x = O.gen_tuples(0, 2000, F, 0)
        got = eng.score_device(torch.from_numpy(x.view(np.int32)).cuda()).cpu().numpy()
        want = O.score(O.Model(m.params, w, m.flines), x, sum_mode=O.SUM_REF_NATIVE)
        assert np.array_equal(_bits(got), _bits(want))
    finally:
        eng.set_option("leaf_domain_check", 1)


def test_empty_tree_shards_score_zero_on_every_rank(eng):
    """ceil(T/G) trees per device leaves trailing devices without trees (T = 9, G = 4 -> 3, 3, 3, 0): such an engine holds
    only EMPTY slots (DTPU.sv:544,760) and returns +0 instead of failing on one rank while the others wait in a collective."""
    import torch

    T, D, F, G = 9, 6, 12, 4
    m = O.gen_model(T, D, F, 1)
    x = O.gen_tuples(0, 1500, F, 1)
    parts = np.stack([_gpu_score(eng, m, x, shard=(g, G)) for g in range(G)])
    assert not parts[3].any() and eng.info().local_trees == 0
    got = eng.chain_sum_device(torch.from_numpy(parts).cuda()).cpu().numpy()
    assert np.array_equal(_bits(got), _bits(O.score(m, x, n_devices=G)))
    with pytest.raises(ddt.DDTError):
        eng.load_model(_params(m), m.wlines, m.flines, 0, T + 1)  # more shards than trees: refused on every rank alike


def test_engine_bookkeeping_num_classes_and_reserve_rows(eng):
    import torch

    m = O.gen_model(300, 8, 32, 0)
    eng.load_model_multiclass(ddt.make_params(300, 8, 32, clusters=1), m.wlines, m.flines, 3, True)
    assert eng.num_classes == 3
    eng.load_model(_params(m), m.wlines, m.flines)
    assert eng.num_classes == 1  # a plain load after a multi-class one resets it
    assert eng.info().variant_name.decode().startswith("q16")
    eng.set_option("reserve_rows", 50_000)  # the rank-quantised path's workspace exists before the first asynchronous call
    x = O.gen_tuples(0, 40_000, 32, 0)
    got = eng.score_device(torch.from_numpy(x.view(np.int32)).cuda()).cpu().numpy()
    assert np.array_equal(_bits(got), _bits(O.score_fast(m, x)))
