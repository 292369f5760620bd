"""The REAL host side of libddt -- csrc/ddt_engine.cpp, ddt_model.cpp, ddt_image.cpp, ddt_choice.cpp, ddt_comm.cpp, ddt_codec.cpp, ddt_sparse_host.cpp, compiled unchanged --
run without a GPU against the deferred-execution HIP / RCCL model of tests/mock_hip/ (see test_comm_mock.py) and CPU
stand-ins for the kernels (mock_kernels.cpp) that read what the real kernels read: the packed images the engine uploads, the
tuple lines, the threshold tables and the rank workspace of the rank-quantised path.  Results are held to the oracle bit for
bit, under five stream schedules each.

What this exercises that CPU tests could not reach before: ddt_load_model -> variant choice -> image upload -> launch
arguments; the feeder of ddt_score / ddt_classify (two pinned buffers, two streams, two rank-workspace slots); the class
launches alternating between two streams around a shared pre-pass; back-to-back asynchronous calls; and the tree-sharded
multi-GPU jobs with the real engine on every rank against the oracle's multi-device model.  Two tests remove a dependency
from the engine source and require some schedule to notice.  (The real kernels are held to the oracle by the GPU tests.)"""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import pytest

import ddt
from ddt import _lib
from oracle import oracle as O
from tests.mock_hip.build_lock import build_if_stale

HERE = os.path.dirname(os.path.abspath(__file__))
MOCK = os.path.join(HERE, "mock_hip")
CSRC = os.path.join(os.path.dirname(HERE), "distributed-decisiontrees_amd", "csrc")
vp = C.c_void_p
SOURCES = ["ddt_engine.cpp", "ddt_model.cpp", "ddt_image.cpp", "ddt_choice.cpp", "ddt_comm.cpp", "ddt_codec.cpp", "ddt_sparse_host.cpp"]
SCHEDULES = [(0, 0), (1, 0), (2, 21), (2, 22), (2, 23)]


def _build(name, engine_source=None):
    mode = os.environ.get("DDT_MOCK_SANITIZE", "")   # "1" / "address": AddressSanitizer; "undefined": UBSan (aborts on a finding); see tests/mock_hip/README
    san = (["-fsanitize=undefined", "-fno-sanitize-recover=undefined", "-fno-omit-frame-pointer", "-g"] if mode == "undefined" else
           ["-fsanitize=address", "-fno-omit-frame-pointer", "-g"] if mode else [])
    out = os.path.join(MOCK, name.replace(".so", "_ubsan.so" if mode == "undefined" else "_asan.so") if san else name)
    srcs = [engine_source or os.path.join(CSRC, "ddt_engine.cpp")] + [os.path.join(CSRC, f) for f in SOURCES[1:]] + [os.path.join(MOCK, "mock_kernels.cpp")]
    deps = srcs + [os.path.join(MOCK, "mock_runtime.cpp"), os.path.join(MOCK, "hip", "hip_runtime.h"), os.path.join(MOCK, "rccl", "rccl.h"),
                   os.path.join(CSRC, "ddt_engine_priv.h"), os.path.join(CSRC, "ddt_internal.h")]
    build_if_stale(out, deps, ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-w", *san, "-I" + MOCK, "-I" + CSRC, *srcs])
    L = _lib.bind(C.CDLL(out))
    L.hipSetDevice.argtypes = [C.c_int]
    L.hipStreamCreateWithFlags.argtypes, L.hipStreamSynchronize.argtypes, L.hipStreamDestroy.argtypes = [C.POINTER(vp), C.c_uint], [vp], [vp]
    L.mock_reset.argtypes, L.mock_reset.restype = [C.c_int, C.c_uint64, C.c_int], None
    return L


@pytest.fixture(scope="module")
def mock():
    return _build("libddt_host_mock.so")


def _variant(L, name):
    for i in range(L.ddt_num_variants()):
        b = C.create_string_buffer(64)
        L.ddt_variant_name(i, b, 64)
        if b.value.decode() == name:
            return i
    raise KeyError(name)


def _engine(L, dev=0):
    e = vp()
    assert L.ddt_create(C.byref(e), dev) == 0
    return e


def _stream(L):
    s = vp()
    assert L.hipStreamCreateWithFlags(C.byref(s), 1) == 0
    return s


def _bits(a):
    return np.ascontiguousarray(a).view(np.uint32)


def _load(L, e, m, params, variant=None, shard=(0, 1)):
    assert L.ddt_set_option(e, b"variant", -1 if variant is None else _variant(L, variant)) == 0
    rc = L.ddt_load_model_shard(e, C.byref(params), m.wlines.ctypes.data, m.wlines.size // 4, m.flines.ctypes.data, m.flines.size // 8, *shard)
    assert rc == 0, L.ddt_last_error(e)


@pytest.mark.parametrize("policy,seed", SCHEDULES[:3])
@pytest.mark.parametrize("T,D,F,variant,dist", [(37, 8, 32, "q16_d8_c8_u4_gl", 1), (300, 8, 32, None, 0), (21, 8, 20, "q16_d8_c4_u4", 1),
                                                (100, 6, 28, None, 1), (9, 8, 32, "d8_t1024_r1_c4_u4_dma_f", 1), (12, 11, 40, None, 1),
                                                (64, 4, 16, "d4_t256_r1_c64_u8_dma", 1)])
def test_load_choose_upload_launch(mock, T, D, F, variant, dist, policy, seed):
    """ddt_load_model -> variant -> image -> launch arguments -> scores == oracle, resident tuples, three calls in flight."""
    mock.mock_reset(policy, seed, 8)
    m, x = O.gen_model(T, D, F, dist), O.gen_tuples(0, 1500, F, dist)
    want = O.score(m, x)
    e, s = _engine(mock), _stream(mock)
    _load(mock, e, m, ddt.make_params(T, D, F), variant)
    outs = [np.full(1500, np.nan, np.float32) for _ in range(3)]
    for o in outs:
        assert mock.ddt_score_device(e, x.ctypes.data, 1500, o.ctypes.data, s) == 0
    assert mock.hipStreamSynchronize(s) == 0
    for o in outs:
        assert np.array_equal(_bits(o), _bits(want))
    info = ddt.Info()
    assert mock.ddt_get_info(e, C.byref(info)) == 0
    if variant:
        assert info.variant_name.decode() == variant
    elif T * D >= 640 and D in (6, 8) and F <= 32:
        assert info.variant_name.decode().startswith("q16_")           # the engine's own choice above the break-even
    mock.ddt_destroy(e)


@pytest.mark.parametrize("policy,seed", SCHEDULES)
@pytest.mark.parametrize("variant,feeder_rows", [("q16_d8_c8_u4_gl", 1024), ("q16_d8_c8_u4_gl", 700), ("d8_t1024_r1_c4_u4_dma_f", 333)])
def test_feeder_double_buffering(mock, variant, feeder_rows, policy, seed):
    """ddt_score from host memory: two pinned buffers, two streams, (rank-quantised:) two workspace slots, many chunks."""
    mock.mock_reset(policy, seed, 8)
    T, D, F, n = 40, 8, 32, 6001
    m, x = O.gen_model(T, D, F, 1), O.gen_tuples(0, n, F, 1)
    want = O.score(m, x)
    e = _engine(mock)
    _load(mock, e, m, ddt.make_params(T, D, F), variant)
    assert mock.ddt_set_option(e, b"feeder_rows", feeder_rows) == 0
    for _ in range(2):
        out = np.full(n, np.nan, np.float32)
        assert mock.ddt_score(e, x.ctypes.data, n, out.ctypes.data) == 0, mock.ddt_last_error(e)
        assert np.array_equal(_bits(out), _bits(want))
    st = ddt.Stats()
    assert mock.ddt_get_stats(e, C.byref(st)) == 0 and st.tuples_in == 2 * n and st.tuples_out == 2 * n
    mock.ddt_destroy(e)


@pytest.mark.parametrize("policy,seed", SCHEDULES)
@pytest.mark.parametrize("T,D,F,K,variant", [(100, 8, 32, 10, "q16_d8_c8_u4_gl"), (60, 6, 16, 3, None), (35, 8, 32, 5, "d8_t1024_r1_c4_u4_dma_f"),
                                             # the persistent kernel: every class in ONE launch (classes of equal size: the images stand back
                                             # to back) / one launch per class (37 trees over 5 classes: 8, 8, 7, 7, 7)
                                             (100, 8, 32, 10, "q16_d8_c8_u4_gl_s2_cm_p"), (35, 8, 32, 5, "q16_d8_c8_u4_gl_s2_cm_p"), (37, 8, 32, 5, "q16_d8_c8_u4_gl_s2_cm_p"),
                                             (300, 8, 32, 3, "q16_d8_c8_u4_gl_s2_cm_p"), (264, 8, 32, 2, "q16_d8_c8_u4_gl_s2_cm_p")])
@pytest.mark.parametrize("clusters", [1, 2, 4])
def test_class_launches_on_two_streams(mock, T, D, F, K, variant, policy, seed, clusters):
    """One-vs-all classes: class 0 (+ the shared rank pre-pass) on the caller's stream, odd classes on the engine's own stream.
    clusters > 1 with the one-launch persistent kernel: a class's partly filled PU group sits in the MIDDLE of its cluster-major image
    (10 trees per class, 2 clusters: groups {0}, {1} -> the partial group 1 is last; 100 per class, 2 clusters: group 12 of 13 lands at
    position 6) -- the kernel skips the padding half of the chunk the host names."""
    if clusters > 1 and (variant is None or "_cm" not in variant):
        pytest.skip("cluster-major images only")
    mock.mock_reset(policy, seed, 8)
    n = 2100
    m, x = O.gen_model(T, D, F, 1, clusters=clusters), O.gen_tuples(0, n, F, 1)
    p = ddt.make_params(T, D, F, clusters=clusters)
    labels, cs = O.classify(m, x, K)
    e, s = _engine(mock), _stream(mock)
    assert mock.ddt_set_option(e, b"variant", -1 if variant is None else _variant(mock, variant)) == 0
    assert mock.ddt_load_model_multiclass(e, C.byref(p), m.wlines.ctypes.data, m.wlines.size // 4, m.flines.ctypes.data, m.flines.size // 8, K, 1, 0, 1) == 0
    res = []
    for _ in range(3):                                                   # back to back: the shared workspace must not be overwritten early
        gs, gl = np.full((K, n), np.nan, np.float32), np.full(n, -1, np.int32)
        assert mock.ddt_classify_device(e, x.ctypes.data, n, gs.ctypes.data, gl.ctypes.data, s) == 0
        res.append((gs, gl))
    assert mock.hipStreamSynchronize(s) == 0
    for gs, gl in res:
        assert np.array_equal(_bits(gs), _bits(cs)) and np.array_equal(gl, labels)
    assert mock.ddt_set_option(e, b"feeder_rows", 512) == 0               # and through the feeder: both feeder streams share the class stream
    hl, hs = np.full(n, -1, np.int32), np.full((K, n), np.nan, np.float32)
    assert mock.ddt_classify(e, x.ctypes.data, n, hl.ctypes.data, hs.ctypes.data) == 0
    assert np.array_equal(hl, labels) and np.array_equal(_bits(hs), _bits(cs))
    mock.ddt_destroy(e)


@pytest.mark.parametrize("policy,seed", SCHEDULES[:3])
@pytest.mark.parametrize("G,T,C_", [(2, 64, 1), (4, 300, 2), (8, 1000, 8), (4, 9, 1)])
def test_tree_sharded_job_with_the_real_engine_equals_the_oracles_multi_device_model(mock, G, T, C_, policy, seed):
    """Rank g loads shard g (ddt_load_model_shard), the partial scores are combined by the C++ pipeline: the chain combine adds
    p0 + p1 + ... in the reference's hop order, and so does this model's all-reduce -- both must equal orc_score(n_devices = G)."""
    mock.mock_reset(policy, seed, 8)
    D, F, n = 8, 32, 1300
    m, x = O.gen_model(T, D, F, 1, clusters=C_), O.gen_tuples(0, n, F, 1)
    want = O.score(m, x, n_devices=G)
    p = ddt.make_params(T, D, F, clusters=C_)
    barrier, shared, errors = threading.Barrier(G), {}, []

    def body(r):
        try:
            assert mock.hipSetDevice(r) == 0
            e, s = _engine(mock, r), _stream(mock)
            _load(mock, e, m, p, None, (r, G))
            if r == 0:
                shared["id"] = C.create_string_buffer(128)
                assert mock.ddt_comm_get_unique_id(shared["id"]) == 0
            barrier.wait()
            c = vp()
            info = ddt.Info()
            assert mock.ddt_get_info(e, C.byref(info)) == 0
            alone = info.variant_name.decode()
            assert mock.ddt_comm_create(C.byref(c), e, r, G, shared["id"]) == 0
            # inside a multi-rank job the rank-quantised depth-8 choice becomes the persistent kernel (tiles from a ticket counter: its blocks
            # do not wait for the CUs the collectives' kernels occupy) -- on the image that is already loaded, without re-packing
            assert mock.ddt_get_info(e, C.byref(info)) == 0
            assert info.variant_name.decode() == ("q16_d8_c8_u4_gl_s2_cm_p" if alone == "q16_d8_c8_u4_gl_s2_cm_x" else alone)
            assert mock.ddt_comm_set_option(c, b"chunk_rows", 500) == 0 and mock.ddt_comm_set_option(c, b"taper_min_rows", 32) == 0
            outs = []
            for combine in (1, 0, 1):
                o = np.full(n, np.nan, np.float32)
                assert mock.ddt_score_sharded_device(c, x.ctypes.data, n, o.ctypes.data, combine, s) == 0, mock.ddt_comm_last_error(c)
                outs.append(o)
            assert mock.hipStreamSynchronize(s) == 0
            for o in outs:
                assert np.array_equal(_bits(o), _bits(want)), r
            barrier.wait()
            mock.ddt_comm_destroy(c)
            mock.ddt_destroy(e)
        except BaseException as ex:  # noqa: BLE001
            errors.append((r, repr(ex)))
            barrier.abort()

    th = [threading.Thread(target=body, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join(180) for t in th]
    assert not errors, errors
    assert mock.mock_errors() == 0


@pytest.mark.parametrize("policy,seed", SCHEDULES[:3])
def test_row_sharded_job_with_the_real_engine(mock, policy, seed):
    """Replicas only: every rank holds the whole ensemble and scores its rows; every rank ends up with the reference-order scores."""
    mock.mock_reset(policy, seed, 8)
    G, T, D, F, n = 4, 120, 8, 32, 2307
    m, x = O.gen_model(T, D, F, 1), O.gen_tuples(0, n, F, 1)
    want = O.score(m, x)
    p = ddt.make_params(T, D, F)
    barrier, shared, errors = threading.Barrier(G), {}, []

    def body(r):
        try:
            assert mock.hipSetDevice(r) == 0
            e, s = _engine(mock, r), _stream(mock)
            _load(mock, e, m, p)
            if r == 0:
                shared["id"] = C.create_string_buffer(128)
                assert mock.ddt_comm_get_unique_id(shared["id"]) == 0
            barrier.wait()
            c = vp()
            assert mock.ddt_comm_create(C.byref(c), e, r, G, shared["id"]) == 0
            assert mock.ddt_comm_set_option(c, b"chunk_rows", 400) == 0
            outs = [np.full(n, np.nan, np.float32) for _ in range(2)]
            for o in outs:
                assert mock.ddt_score_rowsharded_device(c, x.ctypes.data, n, o.ctypes.data, s) == 0
            assert mock.hipStreamSynchronize(s) == 0
            for o in outs:
                assert np.array_equal(_bits(o), _bits(want)), r
            barrier.wait()
            mock.ddt_comm_destroy(c)
            mock.ddt_destroy(e)
        except BaseException as ex:  # noqa: BLE001
            errors.append((r, repr(ex)))
            barrier.abort()

    th = [threading.Thread(target=body, args=(r,)) for r in range(G)]
    [t.start() for t in th]
    [t.join(180) for t in th]
    assert not errors, errors
    assert mock.mock_errors() == 0


def test_single_process_group_with_the_real_engine(mock):
    mock.mock_reset(2, 5, 8)
    T, D, F, n, G = 200, 8, 32, 2500, 4
    m, x = O.gen_model(T, D, F, 0), O.gen_tuples(0, n, F, 0)
    p = ddt.make_params(T, D, F)
    g = vp()
    assert mock.ddt_group_create(C.byref(g), G, None) == 0
    assert mock.ddt_group_load_model(g, C.byref(p), m.wlines.ctypes.data, m.wlines.size // 4, m.flines.ctypes.data, m.flines.size // 8) == 0
    out = np.full(n, np.nan, np.float32)
    assert mock.ddt_group_score(g, x.ctypes.data, n, out.ctypes.data, 1) == 0, mock.ddt_group_last_error(g)
    assert np.array_equal(_bits(out), _bits(O.score(m, x, n_devices=G)))
    assert mock.ddt_group_score_rows(g, x.ctypes.data, n, out.ctypes.data) == -4          # the devices hold tree shards
    # the reference's other mode: the whole ensemble on every device, every device scores its rows through its own feeder
    assert mock.ddt_group_load_model_replicated(g, C.byref(p), m.wlines.ctypes.data, m.wlines.size // 4, m.flines.ctypes.data, m.flines.size // 8) == 0
    for rows in (n, 7, 1, 2499):
        out = np.full(rows, np.nan, np.float32)
        assert mock.ddt_group_score_rows(g, x.ctypes.data, rows, out.ctypes.data) == 0, mock.ddt_group_last_error(g)
        assert np.array_equal(_bits(out), _bits(O.score(m, x[:rows])))
    mock.ddt_group_destroy(g)


@pytest.mark.parametrize("policy,seed", SCHEDULES[:3])
@pytest.mark.parametrize("T,depth,F,full,pm,G", [(24, 13, 20, 4, 650, 1), (40, 16, 64, 3, 700, 1), (19, 9, 12, 2, 500, 4), (9, 3, 5, 1, 400, 2),
                                                 (10, 11, 700, 2, 600, 1), (7, 5, 2048, 1, 500, 2)])   # tuples too wide for a feature tile in LDS
def test_sparse_forests(mock, T, depth, F, full, pm, G, policy, seed):
    """ddt_load_model_sparse: validation, re-basing, top / deep image packing, kernel geometry choice -> launch -> the oracle's walk
    of the explicit-children stream; with G > 1 every virtual rank loads its shard and the partial scores are chain-added."""
    mock.mock_reset(policy, seed, 8)
    sp = O.gen_sparse_model(T, depth, F, full, pm, 1)
    n = 900
    x = O.gen_tuples(0, n, F, 1)
    want = O.score_sparse(sp, x, n_devices=G)
    p = ddt.make_sparse_params(T, depth, F)
    mock.ddt_load_model_sparse.argtypes = [vp, C.POINTER(ddt.Params), vp, C.c_size_t, vp, C.c_uint32, C.c_uint32]
    s = _stream(mock)
    parts = []
    for g in range(G):
        e = _engine(mock)
        lines = np.ascontiguousarray(sp.node_lines, np.uint32)
        first = np.ascontiguousarray(sp.first, np.uint64)
        assert mock.ddt_load_model_sparse(e, C.byref(p), lines.ctypes.data, lines.size // 4, first.ctypes.data, g, G) == 0, mock.ddt_last_error(e)
        if F > 600:   # no tile fits: the kernel that gathers its features from global memory
            info = ddt.Info()
            assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode() == "sparse_gf_k6_u8_t256"
        o = np.full(n, np.nan, np.float32)
        assert mock.ddt_score_device(e, x.ctypes.data, n, o.ctypes.data, s) == 0
        assert mock.hipStreamSynchronize(s) == 0
        h = np.full(n, np.nan, np.float32)
        assert mock.ddt_set_option(e, b"feeder_rows", 256) == 0 and mock.ddt_score(e, x.ctypes.data, n, h.ctypes.data) == 0   # and through the feeder
        assert np.array_equal(_bits(h), _bits(o))
        parts.append(o)
        mock.ddt_destroy(e)
    acc = parts[0]
    for q in parts[1:]:
        acc = O.fpadd_bits_batch(_bits(acc), _bits(q)).view(np.float32)                # host -> dev1 -> ... (ResultsCombiner.sv:292-311)
    assert np.array_equal(_bits(acc), _bits(want))


@pytest.mark.parametrize("policy,seed", SCHEDULES[:2])
@pytest.mark.parametrize("T,depth,F,full,pm,G,cmp_mode,sum_mode", [(24, 13, 20, 4, 650, 1, 0, 0), (130, 16, 64, 3, 700, 1, 0, 2), (19, 9, 12, 2, 500, 3, 1, 0),
                                                                   (9, 3, 5, 1, 400, 2, 0, 0), (12, 10, 7, 2, 500, 1, 0, 1), (40, 14, 100, 2, 600, 1, 1, 0)])
def test_sparse_forests_on_32_bit_ranks(mock, T, depth, F, full, pm, G, cmp_mode, sum_mode, policy, seed):
    """The "sparse_r_*" family (csrc/ddt_sparse_r.hip) through the real host side: rank tables -> key blocks + directory
    (pack_rank32_tables), one-word nodes + pair / LEAF records (sparse_pack_host_r), workspace geometry, pre-pass -> scoring order on the
    stream; the stand-in kernels replay the search and the walk on exactly those bytes.  Forced with option sparse_r32 = 1 (the automatic
    rule wants depth >= 13 and two trees per tuple word: the (130, 16, 64) case also checks that it fires by itself)."""
    mock.mock_reset(policy, seed, 8)
    sp = O.gen_sparse_model(T, depth, F, full, pm, 1, cmp_mode=cmp_mode)
    n = 700
    x = O.gen_tuples(0, n, F, 1)
    ref = {0: O.SUM_REF_NATIVE, 1: O.SUM_F64_SEQ, 2: O.SUM_REF_FLOPOCO}[sum_mode]
    p = ddt.make_sparse_params(T, depth, F, cmp_mode=cmp_mode, sum_mode=sum_mode)
    mock.ddt_load_model_sparse.argtypes = [vp, C.POINTER(ddt.Params), vp, C.c_size_t, vp, C.c_uint32, C.c_uint32]
    s = _stream(mock)
    lines = np.ascontiguousarray(sp.node_lines, np.uint32)
    first = np.ascontiguousarray(sp.first, np.uint64)
    info = ddt.Info()
    parts = []
    for g in range(G):
        e = _engine(mock)
        if T >= 128 and depth >= 13 and G == 1:  # the automatic choice
            assert mock.ddt_load_model_sparse(e, C.byref(p), lines.ctypes.data, lines.size // 4, first.ctypes.data, g, G) == 0, mock.ddt_last_error(e)
            assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode().startswith("sparse_r_"), info.variant_name
            assert mock.ddt_set_option(e, b"sparse_r32", 0) == 0
            assert mock.ddt_get_info(e, C.byref(info)) == 0 and not info.variant_name.decode().startswith("sparse_r_")
        assert mock.ddt_set_option(e, b"sparse_r32", 2) == -1
        assert mock.ddt_set_option(e, b"sparse_r32", 1) == 0
        assert mock.ddt_load_model_sparse(e, C.byref(p), lines.ctypes.data, lines.size // 4, first.ctypes.data, g, G) == 0, mock.ddt_last_error(e)
        assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode().startswith("sparse_r_"), info.variant_name
        o = np.full(n, np.nan, np.float32)
        assert mock.ddt_score_device(e, x.ctypes.data, n, o.ctypes.data, s) == 0
        assert mock.hipStreamSynchronize(s) == 0
        h = np.full(n, np.nan, np.float32)
        assert mock.ddt_set_option(e, b"feeder_rows", 256) == 0 and mock.ddt_score(e, x.ctypes.data, n, h.ctypes.data) == 0   # and through the feeder
        assert np.array_equal(_bits(h), _bits(o))
        # switching the family off and on again re-packs the loaded forest and keeps scoring (the workspace geometry changes with it)
        assert mock.ddt_set_option(e, b"sparse_r32", 0) == 0 and mock.ddt_score(e, x.ctypes.data, n, h.ctypes.data) == 0
        assert np.array_equal(_bits(h), _bits(o))
        assert mock.ddt_set_option(e, b"sparse_r32", 1) == 0 and mock.ddt_score(e, x.ctypes.data, n, h.ctypes.data) == 0
        assert np.array_equal(_bits(h), _bits(o))
        parts.append(o)
        mock.ddt_destroy(e)
    if G == 1:
        want = O.score_sparse(sp, x, sum_mode=ref)
        assert np.array_equal(_bits(parts[0]), _bits(want))
    else:
        want = O.score_sparse(sp, x, n_devices=G)
        acc = parts[0]
        for q in parts[1:]:
            acc = O.fpadd_bits_batch(_bits(acc), _bits(q)).view(np.float32)
        assert np.array_equal(_bits(acc), _bits(want))


def test_the_cli_host_program_on_the_model(mock, tmp_path):
    """csrc/ddt_cli.cpp linked against the model build: gen -> score (one engine through the feeder; --shard i --of n; the
    single-process multi-GPU job --devices 4) -> whole result lines equal to the oracle / its multi-device model."""
    cli = os.path.join(MOCK, "ddt_cli_mock")
    so = os.path.join(MOCK, "libddt_host_mock.so")
    src = os.path.join(CSRC, "ddt_cli.cpp")
    build_if_stale(cli, [src, so], ["g++", "-std=c++17", "-O1", "-Wall", src, so, "-Wl,-rpath," + MOCK, "-pthread"])
    pre = str(tmp_path / "job")
    T, D, F, n = 96, 6, 28, 1203
    subprocess.check_call([cli, "gen", "--trees", str(T), "--levels", str(D), "--features", str(F), "--rows", str(n), "--dist", "1",
                           "--devices", "4", "--prefix", pre])
    m, x = O.gen_model(T, D, F, dist=1, clusters=1), O.gen_tuples(0, n, F, dist=1)
    base = [cli, "score", "--csr", pre + ".csr", "--weights", pre + ".weights", "--findex", pre + ".findex", "--tuples", pre + ".tuples", "--out", pre + ".res"]
    out = subprocess.check_output(base + ["--devices", "4", "--combine", "chain"]).decode()
    assert f"scored {n} tuples on 4 device(s)" in out
    res = np.fromfile(pre + ".res", np.float32)
    assert res.size == (n + 3) // 4 * 4 and not res[n:].any()
    assert np.array_equal(_bits(res[:n]), _bits(O.score(m, x, n_devices=4)))
    out = subprocess.check_output(base + ["--devices", "4", "--mode", "rows"]).decode()
    assert "tuples partitioned" in out and "no collective" in out
    assert np.array_equal(_bits(np.fromfile(pre + ".res", np.float32)[:n]), _bits(O.score(m, x)))
    # the two modes composed: 2 row groups x 2 tree shards (chain combine inside a row group = the oracle's 2-device model on every row)
    out = subprocess.check_output(base + ["--devices", "4", "--mode", "hybrid", "--tree-ranks", "2", "--combine", "chain"]).decode()
    assert "2 row group(s) x 2 tree shard(s)" in out
    assert np.array_equal(_bits(np.fromfile(pre + ".res", np.float32)[:n]), _bits(O.score(m, x, n_devices=2)))
    assert subprocess.run(base + ["--devices", "4", "--mode", "hybrid", "--tree-ranks", "3"], capture_output=True).returncode == 1
    parts = []
    for g in range(4):                                           # the same job as four single-engine runs, combined by the oracle's hop adder
        subprocess.check_output(base + ["--shard", str(g), "--of", "4"])
        parts.append(np.fromfile(pre + ".res", np.float32)[:n])
    acc = parts[0]
    for q in parts[1:]:
        acc = O.fpadd_bits_batch(_bits(acc), _bits(q)).view(np.float32)
    assert np.array_equal(_bits(acc), _bits(O.score(m, x, n_devices=4)))


@pytest.mark.parametrize("seed", [1, 2])
def test_random_models_options_and_call_sequences(mock, seed):
    """A trimmed copy of the exploratory fuzz loop (720 scenarios run once, no failure): random models (depth, width, classes,
    compare and summation mode, clusters, shard), random variant and option settings, reloads on the same engine, resident and
    host calls of awkward sizes -- every result equal to the oracle."""
    rng = np.random.default_rng(seed)
    nvar = mock.ddt_num_variants()
    for _ in range(14):
        mock.mock_reset(int(rng.integers(0, 3)), int(rng.integers(0, 1000)), 8)
        e, s = _engine(mock), _stream(mock)
        for _round in range(int(rng.integers(1, 4))):
            D, T, F = int(rng.choice([4, 6, 8, 8, 8, 11, 3])), int(rng.integers(1, 120)), int(rng.choice([5, 16, 28, 32, 32, 40, 64]))
            cmp_mode, sum_mode, Cc = int(rng.integers(0, 2)), int(rng.integers(0, 2)), int(rng.choice([1, 2, 4, 8]))
            K = min(int(rng.choice([1, 1, 1, 2, 3, 7])), T)
            m = O.gen_model(T, D, F, 1, cmp_mode=cmp_mode, clusters=Cc)
            p = ddt.make_params(T, D, F, cmp_mode=cmp_mode, clusters=Cc, sum_mode=sum_mode)
            G = int(rng.choice([1, 1, 2, 3]))
            g = int(rng.integers(0, G))
            v = -1 if rng.random() < 0.5 else int(rng.integers(0, nvar))
            if mock.ddt_set_option(e, b"variant", v) != 0:                # does not fit the model that is still loaded
                v = -1
                assert mock.ddt_set_option(e, b"variant", -1) == 0
            for opt, val in ((b"class_streams", int(rng.integers(0, 2))), (b"q16_fused_prepass", int(rng.integers(0, 2))),
                             (b"q16_grouped_prepass", int(rng.integers(0, 2))), (b"feeder_rows", int(rng.choice([64, 300, 1024, 5000]))),
                             (b"kernel_timing", int(rng.integers(0, 2)))):
                assert mock.ddt_set_option(e, opt, val) == 0
            args = (C.byref(p), m.wlines.ctypes.data, m.wlines.size // 4, m.flines.ctypes.data, m.flines.size // 8)
            if K > 1:
                if G > -(-T // K):
                    G, g = 1, 0
                rc = mock.ddt_load_model_multiclass(e, *args, K, 1, g, G)
            else:
                if G > T:
                    G, g = 1, 0
                rc = mock.ddt_load_model_shard(e, *args, g, G)
            if rc == -5 and v >= 0:
                continue                                                     # the forced variant does not fit this model
            assert rc == 0, (mock.ddt_last_error(e), T, D, F, K, G, g, v)
            sm = O.SUM_REF_FLOPOCO if sum_mode == 0 else O.SUM_F64_SEQ
            for _call in range(int(rng.integers(1, 4))):
                n = int(rng.choice([0, 1, 7, 1023, 1024, 1025, 2500]))
                x = O.gen_tuples(int(rng.integers(0, 1000)), max(n, 1), F, 1)[:n]
                host = rng.random() < 0.5
                ctx = dict(T=T, D=D, F=F, K=K, G=G, g=g, v=v, n=n, host=host, cmp=cmp_mode, sum=sum_mode, C=Cc)
                if K == 1:
                    per = -(-T // G)
                    b0 = min(g * per, T)
                    b1 = min(b0 + per, T)
                    want = O.score_shard(m, x, b0, b1, sum_mode=sm) if (n and b1 > b0) else np.zeros(n, np.float32)
                    out = np.full(n, np.nan, np.float32)
                    if host:
                        rc = mock.ddt_score(e, x.ctypes.data if n else None, n, out.ctypes.data if n else None)
                    else:
                        rc = mock.ddt_score_device(e, x.ctypes.data if n else None, n, out.ctypes.data if n else None, s)
                        assert mock.hipStreamSynchronize(s) == 0
                    assert rc == 0 and np.array_equal(_bits(out), _bits(want)), ctx
                elif n:
                    gs, gl = np.full((K, n), np.nan, np.float32), np.full(n, -1, np.int32)
                    if host:
                        rc = mock.ddt_classify(e, x.ctypes.data, n, gl.ctypes.data, gs.ctypes.data)
                    else:
                        rc = mock.ddt_classify_device(e, x.ctypes.data, n, gs.ctypes.data, gl.ctypes.data, s)
                        assert mock.hipStreamSynchronize(s) == 0
                    assert rc == 0, ctx
                    if G == 1:
                        labels, cs = O.classify(m, x, K, True, sum_mode=sm)
                        assert np.array_equal(_bits(gs), _bits(cs)) and np.array_equal(gl, labels), ctx
        mock.ddt_destroy(e)


def test_failed_changes_leave_the_loaded_model_in_place(mock):
    """A variant that does not fit, an out-of-range variant id, a malformed reload, unknown or absurd options: each is refused with
    an error code and the model that was loaded keeps scoring."""
    mock.mock_reset(0, 0, 8)
    e = _engine(mock)
    m, x = O.gen_model(30, 8, 32, 1), O.gen_tuples(0, 500, 32, 1)
    want, out = O.score(m, x), np.zeros(500, np.float32)
    p = ddt.make_params(30, 8, 32)

    def scores_still_right():
        out[:] = np.nan
        return mock.ddt_score(e, x.ctypes.data, 500, out.ctypes.data) == 0 and np.array_equal(_bits(out), _bits(want))

    assert mock.ddt_score(e, x.ctypes.data, 500, out.ctypes.data) == -4                        # nothing loaded yet
    _load(mock, e, m, p)
    assert scores_still_right()
    assert mock.ddt_set_option(e, b"variant", _variant(mock, "q16_d6_c16_u4")) == -5 and b"does not fit" in mock.ddt_last_error(e)
    assert scores_still_right()
    assert mock.ddt_set_option(e, b"variant", 9999) == -1 and scores_still_right()
    bad = m.flines.copy()
    bad[0] = 40                                                                                  # feature index >= F
    assert mock.ddt_load_model(e, C.byref(p), m.wlines.ctypes.data, m.wlines.size // 4, bad.ctypes.data, bad.size // 8) == -1
    assert scores_still_right()
    for key in (b"feeder_rows", b"feeder_threads", b"reserve_rows", b"leaf_domain_check", b"nonsense"):
        for val in (-5, 0, 1, 1 << 40):
            mock.ddt_set_option(e, key, val)
    assert mock.ddt_set_option(e, b"feeder_rows", 1 << 20) == 0 and scores_still_right()
    mock.ddt_destroy(e)
    mock.ddt_destroy(None)
    mock.ddt_comm_destroy(None)
    mock.ddt_group_destroy(None)


REMOVED = {
    # the odd classes no longer wait for class 0's launch (and the rank pre-pass in front of it) on the caller's stream
    "class_stream_start": ("    if (two && k == 0) HIP_TRY(e, hipStreamWaitEvent(e->class_stream, e->class_ev[0], 0));\n  }\n  if (two) {\n    HIP_TRY(e, hipEventRecord(e->class_ev[1], e->class_stream));",
                           "  }\n  if (two) {\n    HIP_TRY(e, hipEventRecord(e->class_ev[1], e->class_stream));"),
    # the caller's stream no longer waits for the classes that ran on the engine's stream
    "class_stream_end": ("  if (two) {\n    HIP_TRY(e, hipEventRecord(e->class_ev[1], e->class_stream));\n    HIP_TRY(e, hipStreamWaitEvent(s, e->class_ev[1], 0));\n",
                         "  if (two) {\n    HIP_TRY(e, hipEventRecord(e->class_ev[1], e->class_stream));\n"),
    # the feeder refills a pinned buffer without waiting for the chunk that still uses it
    "feeder_drain": ("    if (pending_n[b] && (rc = drain(b))) return rc;\n    // ALL host-to-device copies", "    // ALL host-to-device copies"),
    # a slot's kernels no longer wait for their chunk to arrive over the (separate) copy stream
    "feeder_h2d_event": ("    HIP_TRY(e, hipStreamWaitEvent(e->fs[b], e->fe_in[b], 0));\n", ""),
}


@pytest.mark.skipif(bool(os.environ.get("DDT_MOCK_SANITIZE")), reason="the broken builds race on purpose")
@pytest.mark.parametrize("which", sorted(REMOVED))
def test_the_model_catches_a_missing_dependency_in_the_engine(which):
    needle, repl = REMOVED[which]
    src = open(os.path.join(CSRC, "ddt_engine.cpp")).read()
    assert src.count(needle) == 1, which
    broken, so = os.path.join(MOCK, f"_broken_engine_{which}.cpp"), f"libddt_host_mock_broken_{which}.so"
    open(broken, "w").write(src.replace(needle, repl))
    try:
        bad = _build(so, broken)
        T, D, F, K, n = 60, 8, 32, 6, 3000
        m, x = O.gen_model(T, D, F, 1), O.gen_tuples(0, n, F, 1)
        p = ddt.make_params(T, D, F, clusters=1)
        labels, cs = O.classify(m, x, K)
        wrong = 0
        keep = []   # buffers that operations still queued on the engine's own stream may write: alive until the engine is gone
        for policy, seed in SCHEDULES:
            bad.mock_reset(policy, seed, 8)
            e, s = _engine(bad), _stream(bad)
            assert bad.ddt_set_option(e, b"variant", _variant(bad, "q16_d8_c8_u4_gl")) == 0
            assert bad.ddt_load_model_multiclass(e, C.byref(p), m.wlines.ctypes.data, m.wlines.size // 4, m.flines.ctypes.data, m.flines.size // 8, K, 1, 0, 1) == 0
            ok = True
            if which.startswith("feeder"):
                assert bad.ddt_set_option(e, b"feeder_rows", 256) == 0
                hl, hs = np.full(n, -1, np.int32), np.full((K, n), np.nan, np.float32)
                keep.append((hl, hs))
                assert bad.ddt_classify(e, x.ctypes.data, n, hl.ctypes.data, hs.ctypes.data) == 0
                ok = np.array_equal(hl, labels) and np.array_equal(_bits(hs), _bits(cs))
            else:
                for _ in range(2):
                    gs, gl = np.full((K, n), np.nan, np.float32), np.full(n, -1, np.int32)
                    keep.append((gs, gl))
                    assert bad.ddt_classify_device(e, x.ctypes.data, n, gs.ctypes.data, gl.ctypes.data, s) == 0
                    assert bad.hipStreamSynchronize(s) == 0
                    ok = ok and np.array_equal(_bits(gs), _bits(cs)) and np.array_equal(gl, labels)
            wrong += not ok
            bad.ddt_destroy(e)
        assert wrong > 0, f"no schedule noticed the missing dependency ({which})"
    finally:
        for f in (broken, os.path.join(MOCK, so)):
            if os.path.exists(f):
                os.remove(f)


def test_a_refused_sparse_option_keeps_the_model(mock):
    """a refused sparse_top_levels (no kernel of that K fits) left the engine without a model and with the bad value
    stored; it must behave like a refused "variant": previous setting kept, model still loaded, original error reported."""
    mock.mock_reset(0, 0, 8)
    T, depth, F, n = 12, 12, 20, 700
    sp = O.gen_sparse_model(T, depth, F, 3, 600, 1)
    x = O.gen_tuples(0, n, F, 1)
    want = O.score_sparse(sp, x)
    p = ddt.make_sparse_params(T, depth, F)
    mock.ddt_load_model_sparse.argtypes = [vp, C.POINTER(ddt.Params), vp, C.c_size_t, vp, C.c_uint32, C.c_uint32]
    lines, first = np.ascontiguousarray(sp.node_lines).reshape(-1), np.ascontiguousarray(sp.first)
    e = _engine(mock)
    assert mock.ddt_load_model_sparse(e, C.byref(p), lines.ctypes.data, lines.size // 4, first.ctypes.data, 0, 1) == 0, mock.ddt_last_error(e)
    out = np.zeros(n, np.float32)

    def scores_still_right():
        out[:] = np.nan
        return mock.ddt_score(e, x.ctypes.data, n, out.ctypes.data) == 0 and np.array_equal(_bits(out), _bits(want))

    assert scores_still_right()
    info = ddt.Info()
    assert mock.ddt_get_info(e, C.byref(info)) == 0
    before = info.variant_name.decode()
    assert mock.ddt_set_option(e, b"sparse_top_levels", 7) == -5 and b"no sparse kernel fits" in mock.ddt_last_error(e)   # the CPU model has K = 6, 8, 9 only
    assert scores_still_right()
    assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode() == before
    assert mock.ddt_set_option(e, b"sparse_deep_order", 1) == 0 and scores_still_right()      # an accepted change re-packs and keeps scoring
    assert mock.ddt_set_option(e, b"sparse_top_levels", 6) == 0 and scores_still_right()
    assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode().startswith("sparse_q_k6")
    assert mock.ddt_set_option(e, b"sparse_top_levels", 10) == -5 and scores_still_right()     # refused again: K = 6 stays
    assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode().startswith("sparse_q_k6")
    # rank-quantised kernels off: the fp32-tile kernel of the same K, same scores; a bad value is refused and changes nothing
    # (the fp32-tile path prefers the dense-level-K kernel of the same K where one exists; option sparse_dk = 0: 16-byte level K-1 records)
    assert mock.ddt_set_option(e, b"sparse_q16", 0) == 0 and scores_still_right()
    assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode().startswith("sparse_dk_k6")
    assert mock.ddt_set_option(e, b"sparse_dk", 0) == 0 and scores_still_right()
    assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode().startswith("sparse_k6")
    assert mock.ddt_set_option(e, b"sparse_dk", 2) == -1 and scores_still_right()
    assert mock.ddt_set_option(e, b"sparse_dk", 1) == 0 and scores_still_right()
    assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode().startswith("sparse_dk_k6")
    assert mock.ddt_set_option(e, b"sparse_q16", 2) == -1 and scores_still_right()
    assert mock.ddt_set_option(e, b"sparse_q16", 1) == 0 and scores_still_right()
    assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode().startswith("sparse_q_k6")
    mock.ddt_destroy(e)


@pytest.mark.parametrize("T,clusters", [(1000, 8), (999, 8), (520, 4), (130, 2), (129, 8), (17, 8), (3, 8), (250, 1)])
def test_cluster_major_image_order(mock, T, clusters):
    """`_cm` kernels: the packer stores the PU groups of a cluster together (cluster-major), the launch tells the kernel how many PU
    groups are real, and the scores are the reference-order sums -- checked through the CPU model of the kernel, which reads the
    packed image in that order; the fp64 sum (stream order) refuses such an image; the automatic choice is the cluster-major kernel with
    the pinned read order ("_x") for every cluster count."""
    mock.mock_reset(0, 0, 8)
    D, F, n = 8, 32, 400
    m, x = O.gen_model(T, D, F, 1, clusters=clusters), O.gen_tuples(0, n, F, 1)
    out = np.zeros(n, np.float32)
    e = _engine(mock)
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        p = ddt.make_params(T, D, F, clusters=clusters, sum_mode=sum_mode)
        for name in ("q16_d8_c8_u4_gl_s2_cm", "q16_d8_c8_u4_gl_s2_cm_p"):     # the persistent kernel reads the same image
            _load(mock, e, m, p, name)
            assert mock.ddt_score(e, x.ctypes.data, n, out.ctypes.data) == 0, mock.ddt_last_error(e)
            assert np.array_equal(_bits(out), _bits(O.score(m, x, sum_mode=ref))), (T, clusters, sum_mode, name)
    p64 = ddt.make_params(T, D, F, clusters=clusters, sum_mode=1)
    assert mock.ddt_set_option(e, b"variant", _variant(mock, "q16_d8_c8_u4_gl_s2_cm")) == 0   # (a refused forced variant keeps the loaded model)
    rc = mock.ddt_load_model_shard(e, C.byref(p64), m.wlines.ctypes.data, m.wlines.size // 4, m.flines.ctypes.data, m.flines.size // 8, 0, 1)
    assert rc == -5, rc
    if T * D >= 224 * 8:
        _load(mock, e, m, ddt.make_params(T, D, F, clusters=clusters), None)
        info = ddt.Info()
        assert mock.ddt_get_info(e, C.byref(info)) == 0
        assert info.variant_name.decode() == "q16_d8_c8_u4_gl_s2_cm_x"      # the pinned-read-order kernel: its running total also serves one cluster
    mock.ddt_destroy(e)


@pytest.mark.parametrize("T,F,clusters,parts", [(650, 4, 1, 2), (650, 4, 8, 2), (1100, 4, 4, 2), (1250, 4, 4, 3), (300, 4, 2, 1)])
def test_more_thresholds_than_u16_ranks_hold_is_scored_in_parts(mock, T, F, clusters, parts):
    """u16 ranks stop at 38848 distinct thresholds per feature -- what one block's LDS holds in rank_kernel; 32767 until round 6 -- (the reference allows 8192 nodes x 64 PUs on one feature, DTPU.sv:22,74).
    Beyond that the cluster-major kernels score the ensemble in PARTS -- consecutive chunks of the image with rank tables of their own, a
    pre-pass + a scoring launch per part, the reference-order sum handed from launch to launch (accumulator + running total per tuple):
    bit-exact with the oracle for every cluster count and both adders, through resident and host calls, back to back."""
    mock.mock_reset(2, 5, 8)
    D, n = 8, 700
    m, x = O.gen_model(T, D, F, 0, clusters=clusters), O.gen_tuples(0, n, F, 0)
    x[5, 1] = 0x7FC00000                                                       # one tile with a missing value: the parts' slow images
    e, st = _engine(mock), ddt.Stats()
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        _load(mock, e, m, ddt.make_params(T, D, F, clusters=clusters, sum_mode=sum_mode), None)
        assert mock.ddt_set_option(e, b"q16_cluster_split", 0) == 0     # (an ensemble in ONE part: a batch this small would be cut at its clusters)
        info = ddt.Info()
        assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode() == "q16_d8_c8_u4_gl_s2_cm_x"
        want = O.score(m, x, sum_mode=ref)
        assert mock.ddt_get_stats(e, C.byref(st)) == 0
        before = st.kernel_launches
        outs = [np.full(n, np.nan, np.float32) for _ in range(2)]
        s = _stream(mock)
        for out in outs:                                                         # back to back: the state workspace is reused in stream order
            assert mock.ddt_score_device(e, x.ctypes.data, n, out.ctypes.data, s) == 0, mock.ddt_last_error(e)
        assert mock.hipStreamSynchronize(s) == 0
        for out in outs:
            assert np.array_equal(_bits(out), _bits(want)), (T, clusters, sum_mode)
        assert mock.ddt_get_stats(e, C.byref(st)) == 0 and st.kernel_launches - before == 2 * parts    # one scoring launch per part
        host = np.full(n, np.nan, np.float32)
        assert mock.ddt_set_option(e, b"feeder_rows", 256) == 0
        assert mock.ddt_score(e, x.ctypes.data, n, host.ctypes.data) == 0 and np.array_equal(_bits(host), _bits(want))
    # the fp64 sum (stream order) cannot run on a cluster-major image: such a model falls back to the fp32 tile kernel
    _load(mock, e, m, ddt.make_params(T, D, F, clusters=clusters, sum_mode=1), None)
    info = ddt.Info()
    assert mock.ddt_get_info(e, C.byref(info)) == 0
    assert info.variant_name.decode().startswith("q16_") == (parts == 1)
    mock.ddt_destroy(e)


def test_a_lowered_table_limit_never_fails_a_load(mock):
    """Option "q16_max_table" (A/B, tests): with a limit that ONE PU group of 8 trees exceeds on a feature (8 x 255 nodes on 2 features against 500 keys) the
    rank-quantised kernels do not fit -- the engine takes another kernel and scores the oracle's bits; with a limit the groups fit, it scores in parts; out of
    range: refused, the previous value kept."""
    mock.mock_reset(2, 3, 8)
    T, D, F, n = 240, 8, 2, 900
    m, x = O.gen_model(T, D, F, 0), O.gen_tuples(0, n, F, 0)
    want = O.score_fast(m, x, sum_mode=O.SUM_REF_NATIVE)
    e, info, st = _engine(mock), ddt.Info(), ddt.Stats()
    assert mock.ddt_set_option(e, b"q16_max_table", 38849) != 0 and mock.ddt_set_option(e, b"q16_max_table", 254) != 0
    for limit, q16 in ((500, False), (3000, True), (38848, True)):
        assert mock.ddt_set_option(e, b"q16_max_table", limit) == 0
        _load(mock, e, m, ddt.make_params(T, D, F), None)
        assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode().startswith("q16_") == q16, (limit, info.variant_name)
        out = np.full(n, np.nan, np.float32)
        assert mock.ddt_set_option(e, b"q16_cluster_split", 0) == 0 and mock.ddt_get_stats(e, C.byref(st)) == 0
        before = st.kernel_launches
        assert mock.ddt_score(e, x.ctypes.data, n, out.ctypes.data) == 0, mock.ddt_last_error(e)
        assert np.array_equal(_bits(out), _bits(want)), limit
        assert mock.ddt_get_stats(e, C.byref(st)) == 0
        if limit == 3000:
            assert st.kernel_launches - before >= 10                             # 30 PU groups of ~1020 keys per feature: two groups per part
    mock.ddt_destroy(e)


@pytest.mark.parametrize("policy,seed", SCHEDULES[:3])
def test_registered_host_buffers_skip_the_staging(mock, policy, seed):
    """ddt_host_register: tuples and scores move straight between the caller's (pinned) buffers and the device; same results, many
    chunks through the three feeder slots; a buffer outside the registered range still takes the staged path."""
    mock.mock_reset(policy, seed, 8)
    T, D, F, n = 40, 8, 32, 5003
    m, x = O.gen_model(T, D, F, 1), O.gen_tuples(0, n, F, 1)
    want = O.score(m, x)
    e = _engine(mock)
    _load(mock, e, m, ddt.make_params(T, D, F), "q16_d8_c8_u4_gl_s2")
    assert mock.ddt_set_option(e, b"feeder_rows", 600) == 0
    out = np.full(n, np.nan, np.float32)
    assert mock.ddt_host_register(e, x.ctypes.data, x.nbytes) == 0
    assert mock.ddt_host_register(e, x.ctypes.data, x.nbytes) == -1            # twice
    assert mock.ddt_host_register(e, out.ctypes.data, out.nbytes) == 0
    assert mock.ddt_score(e, x.ctypes.data, n, out.ctypes.data) == 0, mock.ddt_last_error(e)
    assert np.array_equal(_bits(out), _bits(want))
    other = np.full(n, np.nan, np.float32)                                     # unregistered output: drained through the pinned slot
    assert mock.ddt_score(e, x.ctypes.data, n, other.ctypes.data) == 0 and np.array_equal(_bits(other), _bits(want))
    assert mock.ddt_host_unregister(e, x.ctypes.data) == 0 and mock.ddt_host_unregister(e, x.ctypes.data) == -1
    out[:] = np.nan
    assert mock.ddt_score(e, x.ctypes.data, n, out.ctypes.data) == 0 and np.array_equal(_bits(out), _bits(want))   # staged in, direct out
    mock.ddt_destroy(e)                                                        # hands back what is still registered


def test_kernel_timing_ring(mock):
    """kernel_timing: every timed launch takes an event triple from a ring of 64; the counters are folded in when they are read, a full
    ring waits for its oldest launch, a sparse model and a refused call leave the ring consistent."""
    mock.mock_reset(2, 9, 8)
    T, D, F, n = 40, 8, 32, 1100
    m, x = O.gen_model(T, D, F, 1), O.gen_tuples(0, n, F, 1)
    want = O.score(m, x)
    e, s = _engine(mock), _stream(mock)
    st = ddt.Stats()
    for variant in ("q16_d8_c8_u4_gl", "d8_t1024_r1_c4_u4_dma_f"):
        _load(mock, e, m, ddt.make_params(T, D, F), variant)
        assert mock.ddt_get_stats(e, C.byref(st)) == 0
        before = st.timed_launches
        assert mock.ddt_set_option(e, b"kernel_timing", 1) == 0
        outs = [np.full(n, np.nan, np.float32) for _ in range(150)]
        for o in outs:
            assert mock.ddt_score_device(e, x.ctypes.data, n, o.ctypes.data, s) == 0
        assert mock.ddt_score_device(e, None, n, outs[0].ctypes.data, s) != 0          # refused before anything is recorded
        assert mock.ddt_get_stats(e, C.byref(st)) == 0 and st.timed_launches == before + 150
        assert mock.ddt_get_stats(e, C.byref(st)) == 0 and st.timed_launches == before + 150   # nothing is counted twice
        assert st.sum_score_ms >= st.last_score_ms >= 0 and st.sum_prepass_ms >= 0
        assert mock.ddt_set_option(e, b"kernel_timing", 0) == 0
        assert mock.ddt_score_device(e, x.ctypes.data, n, outs[0].ctypes.data, s) == 0
        assert mock.hipStreamSynchronize(s) == 0
        assert mock.ddt_get_stats(e, C.byref(st)) == 0 and st.timed_launches == before + 150
        for o in outs:
            assert np.array_equal(_bits(o), _bits(want))
    mock.ddt_destroy(e)


def test_random_sparse_forests_options_and_call_sequences(mock):
    """A trimmed copy of a fuzz loop run once over 300 seeds without a failure: random sparse forests (depth, width up to 1500 features,
    compare / summation mode, clusters, shard), random sparse_* options incl. refused ones, reloads on one engine, resident and host
    calls of awkward sizes with missing values -- every result equal to the oracle's score of the shard's sub-forest."""
    mock.ddt_load_model_sparse.argtypes = [vp, C.POINTER(ddt.Params), vp, C.c_size_t, vp, C.c_uint32, C.c_uint32]
    seen = set()
    for seed in range(4, 16):
        _sparse_fuzz_round(mock, seed, seen)
    assert {"sparse_gf", "sparse_qd", "sparse_dk", "sparse"} <= seen, seen


def _sparse_fuzz_round(mock, seed, seen):
    rng = np.random.default_rng(seed)
    mock.mock_reset(int(rng.integers(0, 3)), int(rng.integers(0, 1000)), 8)
    e, s = _engine(mock), _stream(mock)
    for _round in range(int(rng.integers(1, 4))):
        T, depth = int(rng.integers(1, 30)), int(rng.integers(1, 15))
        F = int(rng.choice([1, 5, 20, 64, 65, 130, 600, 1500]))
        full, pm, cmp_mode = min(depth, int(rng.integers(0, 6))), int(rng.integers(0, 1000)), int(rng.integers(0, 2))
        sp = O.gen_sparse_model(T, depth, F, full, pm, 1, cmp_mode=cmp_mode)
        sum_mode = int(rng.choice([0, 1, 2]))
        Cc = int(rng.choice([1, 2, 4, 8]))
        p = ddt.make_sparse_params(T, depth, F, sp.params.missing_bits, cmp_mode, Cc, sum_mode)
        spm = O.SparseModel(O.make_sparse_params(T, depth, F, cmp_mode=cmp_mode, clusters=Cc), sp.node_lines, sp.first)
        G = int(rng.choice([1, 1, 2, 3])); G = min(G, T); g = int(rng.integers(0, G))
        opts = {}
        for opt, vals in ((b"sparse_top_levels", [-1, -1, 6, 7, 8, 9, 10]), (b"sparse_dk", [0, 1]), (b"sparse_q16", [0, 1]), (b"sparse_deep_order", [0, 1]),
                          (b"feeder_rows", [64, 300, 5000]), (b"kernel_timing", [0, 1])):
            v = int(rng.choice(vals)); opts[opt] = v
            rc = mock.ddt_set_option(e, opt, v)
            assert rc in (0, -5), (opt, v, rc)
        lines = np.ascontiguousarray(sp.node_lines, np.uint32); first = np.ascontiguousarray(sp.first, np.uint64)
        rc = mock.ddt_load_model_sparse(e, C.byref(p), lines.ctypes.data, lines.size // 4, first.ctypes.data, g, G)
        if rc == -5:
            continue   # forced K does not fit
        assert rc == 0, (seed, mock.ddt_last_error(e), T, depth, F, opts)
        info = ddt.Info(); mock.ddt_get_info(e, C.byref(info))
        for _call in range(int(rng.integers(1, 3))):
            n = int(rng.choice([0, 1, 63, 257, 1024, 1500]))
            x = O.gen_tuples(int(rng.integers(0, 100)), max(n, 1), F, 1, missing_bits=int(sp.params.missing_bits))[:n]
            if n and rng.random() < 0.5:
                x[:: 3, int(rng.integers(0, F))] = sp.params.missing_bits
            per = -(-T // G); b0 = min(g * per, T); b1 = min(b0 + per, T)
            sm = (O.SUM_REF_NATIVE, O.SUM_F64_SEQ, O.SUM_REF_FLOPOCO)[sum_mode]
            if b1 > b0:
                f0, f1 = int(sp.first[b0]), int(sp.first[b1])
                sub = O.SparseModel(O.make_sparse_params(b1 - b0, depth, F, cmp_mode=cmp_mode, clusters=Cc), np.ascontiguousarray(sp.node_lines).reshape(-1, 4)[f0:f1].copy(),
                                    (np.asarray(sp.first[b0:b1 + 1], np.uint64) - np.uint64(f0)))
                want = O.score_sparse(sub, x, sum_mode=sm) if n else np.zeros(0, np.float32)
            else:
                want = np.zeros(n, np.float32)
            out = np.full(n, np.nan, np.float32)
            if rng.random() < 0.5:
                rc = mock.ddt_score(e, x.ctypes.data if n else None, n, out.ctypes.data if n else None)
            else:
                rc = mock.ddt_score_device(e, x.ctypes.data if n else None, n, out.ctypes.data if n else None, s)
                assert mock.hipStreamSynchronize(s) == 0
            ctx = dict(seed=seed, T=T, depth=depth, F=F, G=G, g=g, n=n, sum=sum_mode, C=Cc, cmp=cmp_mode, opts=opts, var=info.variant_name.decode())
            assert rc == 0, ctx
            assert np.array_equal(_bits(out), _bits(want)), ctx
            seen.add(info.variant_name.decode().rsplit("_k", 1)[0])
    mock.ddt_destroy(e)


@pytest.mark.parametrize("T,D,F,clusters,name,parts", [(20, 12, 32, 1, "q16d_d12_k9_c4_u4_cm", 1), (13, 12, 8, 4, "q16d_d12_k9_c4_u4_cm", 1),
                                                        (40, 12, 4, 2, "q16d_d12_k9_c4_u4_cm", 2), (70, 12, 3, 8, "q16d_d12_k9_c4_u4_cm", 3),
                                                        (11, 10, 16, 1, "q16d_d10_k9_c4_u4_cm", 1), (9, 11, 20, 8, "q16d_d11_k8_c8_u4_cm", 1),
                                                        (17, 9, 32, 2, "q16d_d9_k8_c8_u4_cm", 1), (5, 14, 12, 1, "q16d_d14_k9_c4_u4_cm", 1),
                                                        (9, 15, 12, 2, "q16d_d15_k8_c8_u4_cm", 1), (16, 15, 9, 2, "q16d_d15_k8_c8_u4_cm", 2)])
def test_deep_perfect_trees_on_the_deep_kernels(mock, T, D, F, clusters, name, parts):
    """Perfect trees deeper than 8 levels (the reference's own example: 512 x depth 12, profiler/profiler.cpp:32-38) take the deep
    rank-quantised kernels by themselves: K levels as a heap of 4-byte records, then pair / terminal records of 16 bytes per stage
    (csrc/ddt_internal.h).  The host side -- image packing in cluster-major order, the records' own next-block offsets, parts with rank
    tables of their own when a feature carries more than 38848 distinct thresholds, the sum's state between the parts -- against the
    oracle bit for bit, both adders, tiles with and without missing values, resident and host calls."""
    mock.mock_reset(2, 9, 8)
    n = 1300
    m, x = O.gen_model(T, D, F, 0, clusters=clusters), O.gen_tuples(0, n, F, 0)
    x[1100, F - 1] = 0x7FC00000                                                  # the second tile holds a missing value: slow image there
    e, st, info = _engine(mock), ddt.Stats(), ddt.Info()
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        _load(mock, e, m, ddt.make_params(T, D, F, clusters=clusters, sum_mode=sum_mode), None)
        assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode() == name and info.fallback_kernel == 0
        want = O.score_fast(m, x, sum_mode=ref)
        s = _stream(mock)
        # uncut (one block per tile, the sum's state handed from part to part), and cut into runs of PU groups as a batch this small is by itself:
        # every group's sum goes out, the parts need no state, one combine behind the last part (csrc/ddt_deep.hip SPLIT)
        for split, groups, launches in ((0, -1, parts), (-1, -1, parts + 1 if T > 8 else parts), (1, 2, parts + 1 if T > 8 else parts), (1, 1000, parts + 1 if T > 8 else parts)):
            assert mock.ddt_set_option(e, b"q16_cluster_split", split) == 0 and mock.ddt_set_option(e, b"q16_split_groups", groups) == 0
            assert mock.ddt_get_stats(e, C.byref(st)) == 0
            before = st.kernel_launches
            outs = [np.full(n, np.nan, np.float32) for _ in range(2)]
            for out in outs:
                assert mock.ddt_score_device(e, x.ctypes.data, n, out.ctypes.data, s) == 0, mock.ddt_last_error(e)
            assert mock.hipStreamSynchronize(s) == 0
            for out in outs:
                assert np.array_equal(_bits(out), _bits(want)), (T, D, clusters, sum_mode, split, groups)
            assert mock.ddt_get_stats(e, C.byref(st)) == 0 and st.kernel_launches - before == 2 * launches, (split, groups)
        assert mock.ddt_set_option(e, b"q16_cluster_split", -1) == 0 and mock.ddt_set_option(e, b"q16_split_groups", -1) == 0
        host = np.full(n, np.nan, np.float32)
        assert mock.ddt_set_option(e, b"feeder_rows", 512) == 0
        assert mock.ddt_score(e, x.ctypes.data, n, host.ctypes.data) == 0 and np.array_equal(_bits(host), _bits(want))
    # a shard of a tree-sharded job (trees [b, e) of the list, the whole model's cluster count)
    if T >= 16 and D <= 14:
        _load(mock, e, m, ddt.make_params(T, D, F, clusters=clusters), None, shard=(1, 2))
        out = np.full(n, np.nan, np.float32)
        assert mock.ddt_score_device(e, x.ctypes.data, n, out.ctypes.data, None) == 0 and mock.hipDeviceSynchronize() == 0
        assert np.array_equal(_bits(out), _bits(O.score_shard(m, x, (T + 1) // 2, T, sum_mode=O.SUM_REF_NATIVE)))
    # the fp64 sum runs in stream order: not on a cluster-major image.  Round 6: a perfect tree is a sparse tree whose leaves all sit at depth D --
    # the model goes to the sparse-forest kernels (maybe_score_as_sparse) instead of `generic`; with the switch off: `generic`, and the engine says so
    for via_sparse in (1, 0):
        assert mock.ddt_set_option(e, b"generic_via_sparse", via_sparse) == 0
        _load(mock, e, m, ddt.make_params(T, D, F, clusters=clusters, sum_mode=1), None)
        assert mock.ddt_get_info(e, C.byref(info)) == 0
        if via_sparse:
            assert info.variant_name.decode().startswith("sparse_") and info.fallback_kernel == 0 and info.num_levels == D and info.local_trees == T, info.variant_name
        else:
            assert info.variant_name.decode() == "generic" and info.fallback_kernel == 1
        out = np.full(n, np.nan, np.float32)
        assert mock.ddt_score_device(e, x.ctypes.data, n, out.ctypes.data, None) == 0 and mock.hipDeviceSynchronize() == 0
        assert np.array_equal(_bits(out), _bits(O.score(m, x, sum_mode=O.SUM_F64_SEQ)))
    # a forced perfect-tree kernel takes the model back from the sparse path, the automatic choice hands it over again
    assert mock.ddt_set_option(e, b"generic_via_sparse", 1) == 0
    _load(mock, e, m, ddt.make_params(T, D, F, clusters=clusters, sum_mode=1), None)
    assert mock.ddt_set_option(e, b"variant", _variant(mock, "generic")) == 0 and mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode() == "generic"
    assert mock.ddt_score_device(e, x.ctypes.data, n, out.ctypes.data, None) == 0 and mock.hipDeviceSynchronize() == 0
    assert np.array_equal(_bits(out), _bits(O.score(m, x, sum_mode=O.SUM_F64_SEQ)))
    assert mock.ddt_set_option(e, b"variant", -1) == 0 and mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode().startswith("sparse_")
    assert mock.ddt_score_device(e, x.ctypes.data, n, out.ctypes.data, None) == 0 and mock.hipDeviceSynchronize() == 0
    assert np.array_equal(_bits(out), _bits(O.score(m, x, sum_mode=O.SUM_F64_SEQ)))
    mock.ddt_destroy(e)


@pytest.mark.parametrize("T,D,F,clusters,sum_mode,name", [(230, 8, 64, 2, 0, "q16w_d8_c8_u4_gl_s2_cm_x"), (240, 8, 33, 1, 2, "q16w_d8_c8_u4_gl_s2_cm_x"),
                                                           (226, 8, 50, 4, 1, "q16w_d8_c8_u4_gl"), (9, 12, 64, 1, 0, "q16dw_d12_k9_c4_u4_cm"), (12, 10, 37, 2, 2, "q16dw_d10_k9_c4_u4_cm"),
                                                           (40, 8, 64, 1, 0, None), (230, 8, 72, 1, 0, None)])
def test_tuples_of_33_to_64_words_take_the_wide_rank_quantised_kernels(mock, T, D, F, clusters, sum_mode, name):
    """the rank-quantised path stopped at 32 tuple words (1000 x d8 x 33 features fell to the fp32 tile kernel, deep trees to the
    generic one).  The wide kernels' records carry half the row offset; the host side (choice, image, transpose + rank pre-pass tables for
    up to 64 words) against the oracle.  Small ensembles and tuples beyond 64 words stay where they were."""
    mock.mock_reset(2, 3, 8)
    n = 1200
    m, x = O.gen_model(T, D, F, 1, clusters=clusters), O.gen_tuples(0, n, F, 1)
    ref = {0: O.SUM_REF_NATIVE, 1: O.SUM_F64_SEQ, 2: O.SUM_REF_FLOPOCO}[sum_mode]
    want = O.score_fast(m, x, sum_mode=ref) if sum_mode != 1 else O.score(m, x, sum_mode=ref)
    e, info = _engine(mock), ddt.Info()
    _load(mock, e, m, ddt.make_params(T, D, F, clusters=clusters, sum_mode=sum_mode), None)
    assert mock.ddt_get_info(e, C.byref(info)) == 0
    if name is not None:
        assert info.variant_name.decode() == name, info.variant_name
    else:
        assert not info.variant_name.decode().startswith("q16")
    s = _stream(mock)
    outs = [np.full(n, np.nan, np.float32) for _ in range(2)]
    for out in outs:
        assert mock.ddt_score_device(e, x.ctypes.data, n, out.ctypes.data, s) == 0, mock.ddt_last_error(e)
    assert mock.hipStreamSynchronize(s) == 0
    for out in outs:
        assert np.array_equal(_bits(out), _bits(want))
    host = np.full(n, np.nan, np.float32)
    assert mock.ddt_set_option(e, b"feeder_rows", 500) == 0
    assert mock.ddt_score(e, x.ctypes.data, n, host.ctypes.data) == 0 and np.array_equal(_bits(host), _bits(want))
    mock.ddt_destroy(e)


def _widen(m, F_small, F_wide, seed, **kw):
    """the same trees over F_wide features of which only F_small are tested: feature j of the model becomes feature cols[j] (cols[0] = 0, so
    that the lines' padding entries stay what they are)"""
    rng = np.random.default_rng(seed)
    cols = np.concatenate([[0], np.sort(rng.choice(np.arange(1, F_wide), F_small - 1, replace=False))]).astype(np.uint16)
    fl = m.flines.copy()
    fl = (fl & np.uint16(0xF800)) | cols[fl & np.uint16(0x7FF)]
    q = m.params
    return O.Model(O.make_params(q.num_trees, q.num_levels, F_wide, q.missing_bits, q.cmp_mode, q.clusters_per_tuple), m.wlines, fl), cols


@pytest.mark.parametrize("T,D,Fs,Fw,clusters,sum_mode,name", [(9, 12, 60, 200, 1, 0, "q16dw_d12_k9_c4_u4_cm"), (12, 10, 30, 2048, 2, 2, "q16d_d10_k9_c4_u4_cm"),
                                                               (230, 8, 40, 100, 2, 0, "q16w_d8_c8_u4_gl_s2_cm_x"), (260, 8, 20, 68, 4, 0, "q16_d8_c8_u4_gl_s2_cm_x"),
                                                               (9, 12, 65, 200, 1, 0, "sparse_")])
def test_wide_models_that_test_few_features_are_compacted(mock, T, D, Fs, Fw, clusters, sum_mode, name):
    """a perfect-tree model of more than 64 tuple words (the reference takes F <= 2048, DTPU.sv:22-25,628) fell to `generic`
    or to the fp32 tile kernels whatever it tested.  With at most 64 DISTINCT features in its nodes it now runs on the rank-quantised kernels:
    the pre-pass's transpose gathers those columns (Q16Aux::fmap), tables / tiles / records see the compact width.  65 used features: not compacted -- the sparse-forest kernels (maybe_score_as_sparse)."""
    mock.mock_reset(2, 4, 8)
    n = 1100
    m, cols = _widen(O.gen_model(T, D, Fs, 1, clusters=clusters), Fs, Fw, 5)
    x = O.gen_tuples(0, n, Fw, 1)
    x[::9, cols[1]] = m.params.missing_bits      # missing values on a tested column ...
    x[::7, (int(cols[1]) + 1) % Fw] = m.params.missing_bits if (int(cols[1]) + 1) % Fw not in cols else x[::7, (int(cols[1]) + 1) % Fw]   # ... and on one no node reads
    ref = {0: O.SUM_REF_NATIVE, 2: O.SUM_REF_FLOPOCO}[sum_mode]
    want = O.score_fast(m, x, sum_mode=ref)
    e, info = _engine(mock), ddt.Info()
    _load(mock, e, m, ddt.make_params(T, D, Fw, clusters=clusters, sum_mode=sum_mode), None)
    assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode().startswith(name), info.variant_name   # (65 tested features: the sparse path)
    assert info.tuple_words == (Fw + 3) // 4 * 4 and info.fallback_kernel == 0
    s = _stream(mock)
    outs = [np.full(n, np.nan, np.float32) for _ in range(2)]
    for out in outs:
        assert mock.ddt_score_device(e, x.ctypes.data, n, out.ctypes.data, s) == 0, mock.ddt_last_error(e)
    assert mock.hipStreamSynchronize(s) == 0
    for out in outs:
        assert np.array_equal(_bits(out), _bits(want))
    host = np.full(n, np.nan, np.float32)
    assert mock.ddt_set_option(e, b"feeder_rows", 500) == 0
    assert mock.ddt_score(e, x.ctypes.data, n, host.ctypes.data) == 0 and np.array_equal(_bits(host), _bits(want))
    # a shard of a tree-sharded job compacts the features ITS trees test; one-vs-all classes share one column map
    if T >= 12:
        _load(mock, e, m, ddt.make_params(T, D, Fw, clusters=clusters), None, shard=(1, 2))
        out = np.full(n, np.nan, np.float32)
        assert mock.ddt_score_device(e, x.ctypes.data, n, out.ctypes.data, None) == 0 and mock.hipDeviceSynchronize() == 0
        assert np.array_equal(_bits(out), _bits(O.score_shard(m, x, (T + 1) // 2, T, sum_mode=O.SUM_REF_NATIVE)))
        K = 3
        pk = ddt.make_params(T, D, Fw, clusters=1)
        mk = O.Model(O.make_params(T, D, Fw, clusters=1), m.wlines, m.flines)
        assert mock.ddt_load_model_multiclass(e, C.byref(pk), m.wlines.ctypes.data, m.wlines.size // 4, m.flines.ctypes.data, m.flines.size // 8, K, 1, 0, 1) == 0, mock.ddt_last_error(e)
        wl, wcs = O.classify_fast(mk, x, K, True)
        gl, gs = np.full(n, -1, np.int32), np.full((K, n), np.nan, np.float32)
        assert mock.ddt_classify_device(e, x.ctypes.data, n, gs.ctypes.data, gl.ctypes.data, None) == 0 and mock.hipDeviceSynchronize() == 0
        assert np.array_equal(gl, wl) and np.array_equal(_bits(gs), _bits(wcs))
    # the A/B switch: off, the model is where it was before round 6 -- and still right
    assert mock.ddt_set_option(e, b"feature_compaction", 0) == 0
    _load(mock, e, m, ddt.make_params(T, D, Fw, clusters=clusters, sum_mode=sum_mode), None)
    assert mock.ddt_get_info(e, C.byref(info)) == 0 and not info.variant_name.decode().startswith("q16"), info.variant_name
    assert mock.ddt_score(e, x.ctypes.data, n, host.ctypes.data) == 0 and np.array_equal(_bits(host), _bits(want))
    mock.ddt_destroy(e)


@pytest.mark.parametrize("T,K,F,clusters", [(60, 3, 3, 1), (120, 2, 3, 2), (66, 3, 4, 4)])
def test_multiclass_deep_models_scored_in_parts(mock, T, K, F, clusters):
    """The classes of a one-vs-all model share one set of rank tables; when those exceed the u16 ranks (here 512-tree-like threshold counts on 3-4
    features) every class is cut into parts of its own -- a class may come out as ONE part with tables of its own -- and the classes run one
    after the other on one stream (every part rewrites the batch's rank workspace)."""
    mock.mock_reset(2, 4, 8)
    D, n = 12, 900
    m, x = O.gen_model(T, D, F, 0, clusters=clusters), O.gen_tuples(0, n, F, 0)
    x[7, 0] = 0x7FC00000
    labels, cs = O.classify_fast(m, x, K, True)
    e, s, info = _engine(mock), _stream(mock), ddt.Info()
    p = ddt.make_params(T, D, F, clusters=clusters)
    assert mock.ddt_load_model_multiclass(e, C.byref(p), m.wlines.ctypes.data, m.wlines.size // 4, m.flines.ctypes.data, m.flines.size // 8, K, 1, 0, 1) == 0, mock.ddt_last_error(e)
    assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode() == "q16d_d12_k9_c4_u4_cm"
    res = []
    for _ in range(2):
        gs, gl = np.full((K, n), np.nan, np.float32), np.full(n, -1, np.int32)
        assert mock.ddt_classify_device(e, x.ctypes.data, n, gs.ctypes.data, gl.ctypes.data, s) == 0, mock.ddt_last_error(e)
        res.append((gs, gl))
    assert mock.hipStreamSynchronize(s) == 0
    for gs, gl in res:
        assert np.array_equal(_bits(gs), _bits(cs)) and np.array_equal(gl, labels)
    mock.ddt_destroy(e)


def test_a_pu_group_beyond_the_u16_ranks_falls_back_and_says_so(mock):
    """9 trees of depth 15 on 4 features: one PU group of 8 trees carries ~65 k thresholds per feature -- no part of it fits u16 ranks.  The deep
    kernel must not be chosen (the load used to fail with 'not supported' while planning the parts); the generic kernel scores it, flagged."""
    mock.mock_reset(2, 1, 8)
    T, D, F, n = 9, 15, 4, 300
    m, x = O.gen_model(T, D, F, 0), O.gen_tuples(0, n, F, 0)
    e, info = _engine(mock), ddt.Info()
    for via_sparse in (0, 1):   # (round 6: by default such a model goes on to the sparse-forest kernels, which take any number of thresholds)
        assert mock.ddt_set_option(e, b"generic_via_sparse", via_sparse) == 0
        _load(mock, e, m, ddt.make_params(T, D, F), None)
        assert mock.ddt_get_info(e, C.byref(info)) == 0
        if via_sparse:
            assert info.variant_name.decode().startswith("sparse_") and info.fallback_kernel == 0, info.variant_name
        else:
            assert info.variant_name.decode() == "generic" and info.fallback_kernel == 1
        out = np.full(n, np.nan, np.float32)
        assert mock.ddt_score_device(e, x.ctypes.data, n, out.ctypes.data, None) == 0 and mock.hipDeviceSynchronize() == 0
        assert np.array_equal(_bits(out), _bits(O.score_fast(m, x)))
    mock.ddt_destroy(e)


@pytest.mark.parametrize("T,depth,F,full,pm,dp,name", [(20, 14, 64, 11, 700, -1, "sparse_dp_k8_u8_t256"), (20, 14, 64, 9, 150, -1, "sparse_dm1_k8_u8_t256"),
                                                       (20, 14, 64, 9, 300, 1, "sparse_dp_k8_u8_t256"), (12, 9, 64, 3, 500, 1, "sparse_dp_k8_u8_t256"),
                                                       (20, 14, 64, 11, 700, 0, "sparse_dm1_k8_u8_t256")])
def test_sparse_forests_with_dense_pair_records(mock, T, depth, F, full, pm, dp, name):
    """Round 5: the two levels below the top image as ONE block of 16-byte pair records (a gather decides two levels), the dense block at level
    K + 2 (option sparse_dp: automatic when the forest fills the three levels, or forced -- then also on a forest shallower than the blocks):
    choice, packing and the walk against the oracle, with missing values; the feeder."""
    mock.mock_reset(2, 7, 8)
    sp = O.gen_sparse_model(T, depth, F, full, pm, 1)
    n = 500
    x = O.gen_tuples(0, n, F, 1)
    x[::5, 1] = sp.params.missing_bits
    x[::7, F - 1] = sp.params.missing_bits
    want = O.score_sparse(sp, x)
    p = ddt.make_sparse_params(T, depth, F)
    mock.ddt_load_model_sparse.argtypes = [vp, C.POINTER(ddt.Params), vp, C.c_size_t, vp, C.c_uint32, C.c_uint32]
    e, info = _engine(mock), ddt.Info()
    assert mock.ddt_set_option(e, b"sparse_dp", 2) < 0
    assert mock.ddt_set_option(e, b"sparse_dp", dp) == 0 and mock.ddt_set_option(e, b"sparse_q16", 0) == 0
    lines, first = np.ascontiguousarray(sp.node_lines, np.uint32), np.ascontiguousarray(sp.first, np.uint64)
    assert mock.ddt_load_model_sparse(e, C.byref(p), lines.ctypes.data, lines.size // 4, first.ctypes.data, 0, 1) == 0, mock.ddt_last_error(e)
    assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode() == name, info.variant_name
    o = np.full(n, np.nan, np.float32)
    assert mock.ddt_score_device(e, x.ctypes.data, n, o.ctypes.data, None) == 0 and mock.hipDeviceSynchronize() == 0
    assert np.array_equal(_bits(o), _bits(want))
    h = np.full(n, np.nan, np.float32)
    assert mock.ddt_set_option(e, b"feeder_rows", 256) == 0 and mock.ddt_score(e, x.ctypes.data, n, h.ctypes.data) == 0
    assert np.array_equal(_bits(h), _bits(want))
    mock.ddt_destroy(e)


@pytest.mark.parametrize("dp,name", [(-1, "sparse_qp_k8_u8_t1024"), (0, "sparse_qd_k8_u8_t1024")])
def test_rank_quantised_sparse_forests_with_dense_pair_records(mock, dp, name):
    """... and the rank-quantised family (thresholds -> ranks, the u16 tiles of the q16 pre-pass): the same pair records with ranks as keys"""
    mock.mock_reset(2, 7, 8)
    T, depth, F = 20, 14, 64
    sp = O.gen_sparse_model(T, depth, F, 11, 700, 1)
    n = 1300
    x = O.gen_tuples(0, n, F, 1)
    x[::5, 1] = sp.params.missing_bits
    want = O.score_sparse(sp, x)
    p = ddt.make_sparse_params(T, depth, F)
    mock.ddt_load_model_sparse.argtypes = [vp, C.POINTER(ddt.Params), vp, C.c_size_t, vp, C.c_uint32, C.c_uint32]
    e, info = _engine(mock), ddt.Info()
    assert mock.ddt_set_option(e, b"sparse_dp", dp) == 0
    lines, first = np.ascontiguousarray(sp.node_lines, np.uint32), np.ascontiguousarray(sp.first, np.uint64)
    assert mock.ddt_load_model_sparse(e, C.byref(p), lines.ctypes.data, lines.size // 4, first.ctypes.data, 0, 1) == 0, mock.ddt_last_error(e)
    assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode() == name, info.variant_name
    o = np.full(n, np.nan, np.float32)
    assert mock.ddt_score_device(e, x.ctypes.data, n, o.ctypes.data, None) == 0 and mock.hipDeviceSynchronize() == 0
    assert np.array_equal(_bits(o), _bits(want))
    mock.ddt_destroy(e)


@pytest.mark.parametrize("T,depth,F,full,pm,dm,name", [(20, 14, 64, 10, 700, -1, "sparse_dm1_k8_u8_t256"), (20, 14, 64, 9, 400, -1, "sparse_dm1_k8_u8_t256"),
                                                       (20, 14, 64, 8, 300, -1, "sparse_dk_k8_u8_t256"), (20, 14, 64, 10, 700, 2, "sparse_dm2_k8_u8_t256"),
                                                       (20, 14, 64, 10, 700, 0, "sparse_dk_k8_u8_t256"), (12, 9, 64, 3, 500, 2, "sparse_dm2_k8_u8_t256")])
def test_sparse_forests_with_dense_mid_levels(mock, T, depth, F, full, pm, dm, name):
    """Round 5: the levels right below the top image as 8-byte heap records when the forest fills them (option sparse_dm: automatic by the
    levels' fill, or forced -- then also on a forest shallower than the dense block, which is all padding there): choice, packing (padding
    under early leaves, EMPTY slots' dummy heap) and the walk against the oracle; missing values; the feeder."""
    mock.mock_reset(2, 7, 8)
    sp = O.gen_sparse_model(T, depth, F, full, pm, 1)
    n = 700
    x = O.gen_tuples(0, n, F, 1)
    want = O.score_sparse(sp, x)
    p = ddt.make_sparse_params(T, depth, F)
    mock.ddt_load_model_sparse.argtypes = [vp, C.POINTER(ddt.Params), vp, C.c_size_t, vp, C.c_uint32, C.c_uint32]
    e, info = _engine(mock), ddt.Info()
    assert mock.ddt_set_option(e, b"sparse_dm", dm) == 0 and mock.ddt_set_option(e, b"sparse_q16", 0) == 0   # (a forest this small would fit u16 ranks)
    assert mock.ddt_set_option(e, b"sparse_dp", 0) == 0                                                        # (... and the pair records would take the fuller ones)
    lines, first = np.ascontiguousarray(sp.node_lines, np.uint32), np.ascontiguousarray(sp.first, np.uint64)
    assert mock.ddt_load_model_sparse(e, C.byref(p), lines.ctypes.data, lines.size // 4, first.ctypes.data, 0, 1) == 0, mock.ddt_last_error(e)
    assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode() == name, info.variant_name
    o = np.full(n, np.nan, np.float32)
    assert mock.ddt_score_device(e, x.ctypes.data, n, o.ctypes.data, None) == 0 and mock.hipDeviceSynchronize() == 0
    assert np.array_equal(_bits(o), _bits(want))
    h = np.full(n, np.nan, np.float32)
    assert mock.ddt_set_option(e, b"feeder_rows", 256) == 0 and mock.ddt_score(e, x.ctypes.data, n, h.ctypes.data) == 0
    assert np.array_equal(_bits(h), _bits(want))
    mock.ddt_destroy(e)


@pytest.mark.parametrize("T,D,clusters,name", [(300, 6, 4, "q16_d6_c16_u4_s2"), (1000, 6, 8, "q16_d6_c16_u4_s2"), (203, 7, 2, "q16_d7_c8_u4_s2"), (90, 5, 1, "q16_d5_c32_u4_s2"),
                                                (500, 4, 4, "q16_d4_c64_u8"), (1030, 3, 8, "q16_d3_c128_u8"), (230, 8, 2, "q16w_d8_c8_u4_gl_s2_cm_x")])
def test_small_batch_is_cut_into_runs_of_groups_on_every_depth(mock, T, D, clusters, name):
    """The cut launch on the kernels whose images are in STREAM order (depths 3-7: group g belongs to cluster g mod C, Core.sv:291-316) and on the wide
    depth-8 kernel: a slice is a run of chunks (2 ... 16 PU groups each), every group's sum goes out at the group's place in the image, and the
    combine picks every C-th one for a cluster's chain -- the oracle's bits in both adders, EMPTY padding behind the last real tree included."""
    mock.mock_reset(0, 5, 8)
    F = 40 if name.startswith("q16w") else 10
    m = O.gen_model(T, D, F, 1, clusters=clusters)
    e, st, s = _engine(mock), ddt.Stats(), _stream(mock)
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        _load(mock, e, m, ddt.make_params(T, D, F, clusters=clusters, sum_mode=sum_mode), name)
        for n in (5, 1500):
            x = O.gen_tuples(4, n, F, 1)
            if n > 1100:
                x[1100, 1] = 0x7FC00000
            want = O.score(m, x, sum_mode=ref)
            chunks = -(-((T + 7) // 8) // (mock_chunk_trees(name) // 8))
            for groups in (-1, 2, 3, 1000):
                assert mock.ddt_set_option(e, b"q16_cluster_split", 1) == 0 and mock.ddt_set_option(e, b"q16_split_groups", groups) == 0
                assert mock.ddt_get_stats(e, C.byref(st)) == 0
                before = st.kernel_launches
                out = np.full(n, np.nan, np.float32)
                assert mock.ddt_score_device(e, x.ctypes.data, n, out.ctypes.data, s) == 0, mock.ddt_last_error(e)
                assert mock.hipStreamSynchronize(s) == 0
                assert np.array_equal(_bits(out), _bits(want)), (name, T, clusters, sum_mode, n, groups)
                assert mock.ddt_get_stats(e, C.byref(st)) == 0 and st.kernel_launches - before == (2 if chunks > 1 else 1), (groups, chunks)
    mock.ddt_destroy(e)


def mock_chunk_trees(name):
    return int(name.split("_c")[1].split("_")[0])


@pytest.mark.parametrize("T,clusters", [(300, 8), (300, 2), (125, 8), (230, 4), (224, 8), (1000, 8), (229, 1)])
def test_small_batch_is_cut_at_the_clusters(mock, T, clusters):
    """A batch of a few tiles on the plain depth-8 cluster-major kernel: one block per (tile, SLICE of the image) instead of one per tile -- a
    slice is a cluster (its accumulator goes out) or a run of PU groups (every group's sum goes out) -- then the adds in the reference's order
    (FPAggregator.v:79-131: per cluster acc <- x + acc; Core.sv:486-541: the clusters added in order afterwards) -- bit-exact with the oracle and with the uncut launch in both adders, with a missing value in one tile, for row counts that are
    no whole tiles, back to back on one stream; the automatic rule follows the tile limit; one cluster = nothing to cut."""
    mock.mock_reset(1, 11, 8)
    D, F = 8, 12
    m = O.gen_model(T, D, F, 1, clusters=clusters)
    e, st, s = _engine(mock), ddt.Stats(), _stream(mock)
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        _load(mock, e, m, ddt.make_params(T, D, F, clusters=clusters, sum_mode=sum_mode), "q16_d8_c8_u4_gl_s2_cm_x")
        for n in (1, 700, 1024, 2500):
            x = O.gen_tuples(3, n, F, 1)
            if n > 1100:
                x[1100, 2] = 0x7FC00000                                             # the second tile takes the slow image
            want = O.score(m, x, sum_mode=ref)
            got = {}
            cut = 2 if (T + 7) // 8 > 1 else 1
            for split, groups, launches in ((0, -1, 1), (1, 0, 2 if clusters > 1 else 1), (1, -1, cut), (-1, 5, cut), (-1, 1000, cut)):
                assert mock.ddt_set_option(e, b"q16_cluster_split", split) == 0   # groups: 0 = the slices are the clusters, -1 = automatic (runs of PU
                assert mock.ddt_set_option(e, b"q16_split_groups", groups) == 0   # groups for batches this small), k = k slices
                assert mock.ddt_get_stats(e, C.byref(st)) == 0
                before = st.kernel_launches
                outs = [np.full(n, np.nan, np.float32) for _ in range(2)]
                for out in outs:                                                     # back to back: the partial sums' workspace is reused in stream order
                    assert mock.ddt_score_device(e, x.ctypes.data, n, out.ctypes.data, s) == 0, mock.ddt_last_error(e)
                assert mock.hipStreamSynchronize(s) == 0
                for out in outs:
                    assert np.array_equal(_bits(out), _bits(want)), (T, clusters, sum_mode, n, split)
                assert mock.ddt_get_stats(e, C.byref(st)) == 0 and st.kernel_launches - before == 2 * launches, (split, n)
                got[(split, groups)] = outs[0]
            assert all(np.array_equal(_bits(got[(0, -1)]), _bits(g)) for g in got.values())
        # the automatic rule: batches of up to q16_split_max_tiles tiles
        assert mock.ddt_set_option(e, b"q16_cluster_split", -1) == 0 and mock.ddt_set_option(e, b"q16_split_max_tiles", 2) == 0
        assert mock.ddt_set_option(e, b"q16_split_groups", 0) == 0
        for n, cut in ((2048, True), (2049, False)):
            x = O.gen_tuples(5, n, F, 1)
            out = np.full(n, np.nan, np.float32)
            assert mock.ddt_get_stats(e, C.byref(st)) == 0
            before = st.kernel_launches
            assert mock.ddt_score_device(e, x.ctypes.data, n, out.ctypes.data, s) == 0 and mock.hipStreamSynchronize(s) == 0
            assert np.array_equal(_bits(out), _bits(O.score(m, x, sum_mode=ref)))
            assert mock.ddt_get_stats(e, C.byref(st)) == 0 and st.kernel_launches - before == (2 if cut and clusters > 1 else 1)
        assert mock.ddt_set_option(e, b"q16_split_max_tiles", 384) == 0
    # host buffers (the feeder's slots have workspaces of their own)
    assert mock.ddt_set_option(e, b"q16_cluster_split", 1) == 0 and mock.ddt_set_option(e, b"q16_split_groups", -1) == 0
    x = O.gen_tuples(9, 3000, F, 1)
    out = np.full(3000, np.nan, np.float32)
    assert mock.ddt_score(e, x.ctypes.data, 3000, out.ctypes.data) == 0, mock.ddt_last_error(e)
    assert np.array_equal(_bits(out), _bits(O.score(m, x, sum_mode=O.SUM_REF_FLOPOCO)))
    mock.ddt_destroy(e)


@pytest.mark.parametrize("T,K,clusters", [(1000, 10, 1), (300, 3, 2), (48, 3, 1), (210, 7, 4)])
def test_small_multiclass_batch_is_cut_over_all_classes(mock, T, K, clusters):
    """The one-launch multi-class kernel ("_p": ONE block per tile walks every class) on a batch of a few tiles: the plain kernel's cut form over the
    same image -- the classes stand back to back in it -- with a partial sum per PU group, the combine per class in the reference's order, then the
    argmax: sums and labels equal the uncut launch's and the oracle's bit for bit, in both adders, also for a caller that asks for labels only."""
    mock.mock_reset(1, 3, 8)
    D, F = 8, 12
    m = O.gen_model(T, D, F, 1, clusters=clusters)
    e, st, s = _engine(mock), ddt.Stats(), _stream(mock)
    null = C.c_void_p(None)
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        p = ddt.make_params(T, D, F, clusters=clusters, sum_mode=sum_mode)
        assert mock.ddt_set_option(e, b"variant", _variant(mock, "q16_d8_c8_u4_gl_s2_cm_p")) == 0
        assert mock.ddt_load_model_multiclass(e, C.byref(p), m.wlines.ctypes.data, m.wlines.size // 4, m.flines.ctypes.data, m.flines.size // 8, K, 1, 0, 1) == 0
        for n in (3, 2500):
            x = O.gen_tuples(2, n, F, 1)
            labels, cs = O.classify(m, x, K, sum_mode=ref)
            for split, groups, launches in ((0, -1, 1), (-1, -1, 3), (1, 4, 3), (1, 60000, 3)):
                assert mock.ddt_set_option(e, b"q16_cluster_split", split) == 0 and mock.ddt_set_option(e, b"q16_split_groups", groups) == 0
                assert mock.ddt_get_stats(e, C.byref(st)) == 0
                before = st.kernel_launches
                gs, gl, gl2 = np.full((K, n), np.nan, np.float32), np.full(n, -1, np.int32), np.full(n, -1, np.int32)
                assert mock.ddt_classify_device(e, x.ctypes.data, n, gs.ctypes.data, gl.ctypes.data, s) == 0, mock.ddt_last_error(e)
                assert mock.ddt_get_stats(e, C.byref(st)) == 0 and st.kernel_launches - before == launches, (split, groups)
                assert mock.hipStreamSynchronize(s) == 0
                assert mock.ddt_classify(e, x.ctypes.data, n, gl2.ctypes.data, null) == 0, mock.ddt_last_error(e)             # host buffers, labels only
                assert np.array_equal(_bits(gs), _bits(cs)) and np.array_equal(gl, labels) and np.array_equal(gl2, labels), (T, K, clusters, sum_mode, n, split, groups)
    mock.ddt_destroy(e)


@pytest.mark.parametrize("T,depth,F,clusters", [(130, 14, 20, 1), (130, 14, 20, 4), (70, 13, 9, 8), (21, 10, 6, 2), (8, 9, 5, 4)])
def test_small_batch_on_the_32_bit_rank_sparse_kernels_is_cut_into_slices_of_C_groups(mock, T, depth, F, clusters):
    """`sparse_r_*` on a batch of a few tiles: grid (tiles, slices of C consecutive PU groups) -- in stream order group g belongs to cluster g mod C
    (Core.sv:291-316), so within a slice every cluster's accumulator takes ONE group's sum: the uncut walk's ring IS the groups' sums, the
    epilogue lets it out and launch_cm_combine runs the adds (FPAggregator.v:79-131, Core.sv:486-541).  Oracle's bits, both adders, cut and uncut."""
    mock.mock_reset(1, 4, 8)
    sp = O.gen_sparse_model(T, depth, F, 3, 650, 1, clusters=clusters)
    mock.ddt_load_model_sparse.argtypes = [vp, C.POINTER(ddt.Params), vp, C.c_size_t, vp, C.c_uint32, C.c_uint32]
    lines, first = np.ascontiguousarray(sp.node_lines, np.uint32), np.ascontiguousarray(sp.first, np.uint64)
    e, s, st, info = _engine(mock), _stream(mock), ddt.Stats(), ddt.Info()
    assert mock.ddt_set_option(e, b"sparse_r32", 1) == 0
    for sum_mode, ref in ((0, O.SUM_REF_NATIVE), (2, O.SUM_REF_FLOPOCO)):
        p = ddt.make_sparse_params(T, depth, F, clusters=clusters, sum_mode=sum_mode)
        assert mock.ddt_load_model_sparse(e, C.byref(p), lines.ctypes.data, lines.size // 4, first.ctypes.data, 0, 1) == 0, mock.ddt_last_error(e)
        assert mock.ddt_get_info(e, C.byref(info)) == 0 and info.variant_name.decode().startswith("sparse_r_")
        cuts = (T + 7) // 8 > clusters
        for n in (3, 900):
            x = O.gen_tuples(6, n, F, 1)
            x[n // 2, 1] = sp.params.missing_bits
            want = O.score_sparse(sp, x, sum_mode=ref)
            for split, launches in ((0, 1), (-1, 2 if cuts else 1), (1, 2 if cuts else 1)):
                assert mock.ddt_set_option(e, b"q16_cluster_split", split) == 0
                assert mock.ddt_get_stats(e, C.byref(st)) == 0
                before = st.kernel_launches
                outs = [np.full(n, np.nan, np.float32) for _ in range(2)]
                for o in outs:
                    assert mock.ddt_score_device(e, x.ctypes.data, n, o.ctypes.data, s) == 0, mock.ddt_last_error(e)
                assert mock.hipStreamSynchronize(s) == 0
                for o in outs:
                    assert np.array_equal(_bits(o), _bits(want)), (T, clusters, sum_mode, n, split)
                assert mock.ddt_get_stats(e, C.byref(st)) == 0 and st.kernel_launches - before == 2 * launches, (split, n)
        assert mock.ddt_set_option(e, b"q16_cluster_split", -1) == 0 and mock.ddt_set_option(e, b"sparse_split_max_tiles", 2) == 0   # the kernel's own tiles (256 / 128 tuples)
        x = O.gen_tuples(7, 1100, F, 1)
        o = np.full(1100, np.nan, np.float32)
        assert mock.ddt_get_stats(e, C.byref(st)) == 0
        before = st.kernel_launches
        assert mock.ddt_score_device(e, x.ctypes.data, 1100, o.ctypes.data, s) == 0 and mock.hipStreamSynchronize(s) == 0
        assert mock.ddt_get_stats(e, C.byref(st)) == 0 and st.kernel_launches - before == 1 and np.array_equal(_bits(o), _bits(O.score_sparse(sp, x, sum_mode=ref)))
        assert mock.ddt_set_option(e, b"sparse_split_max_tiles", 256) == 0
    mock.ddt_destroy(e)
