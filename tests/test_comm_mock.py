"""The REAL multi-GPU pipeline code (csrc/ddt_comm.cpp) with 2 .. 8 ranks, on a machine without GPUs.

csrc/ddt_comm.cpp is compiled unchanged against tests/mock_hip/: a deferred-execution model of HIP streams / events and of
the RCCL collectives (mock_runtime.cpp), and a stand-in engine whose partial scores are integer-valued floats (any summation
order gives the same bits).  Ranks are threads.  Operations run only when a scheduler picks them -- lowest stream id first,
highest first, or seeded random -- honouring nothing but stream order and event dependencies, so that a missing wait in the
pipeline shows up as a wrong result under some schedule (the last test removes one on purpose to prove it does).

What this covers that one GPU cannot: shard arithmetic with G > 1, the all-to-all segment layout of the chain combine, the
[K][n] class layout through the workspace slots, ragged row partitions, the tapered tail, back-to-back calls on reused
slots, the host-buffer form, and the single-process group with one worker thread per device."""
import ctypes as C
import os
import subprocess
import threading

import numpy as np
import pytest

import ddt
from tests.mock_hip.build_lock import build_if_stale

HERE = os.path.dirname(os.path.abspath(__file__))
MOCK = os.path.join(HERE, "mock_hip")
CSRC = os.path.join(os.path.dirname(HERE), "distributed-decisiontrees_amd", "csrc")
vp, sz, u32, u64, i32 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int


def _build(name, comm_source):
    san = ["-fsanitize=address", "-fno-omit-frame-pointer", "-g"] if os.environ.get("DDT_MOCK_SANITIZE") else []   # see tests/mock_hip/README
    out = os.path.join(MOCK, name.replace(".so", "_asan.so") if san else name)
    deps = [comm_source, os.path.join(MOCK, "mock_runtime.cpp"), os.path.join(MOCK, "mock_engine.cpp"), os.path.join(MOCK, "hip", "hip_runtime.h"),
            os.path.join(MOCK, "rccl", "rccl.h"),
            os.path.join(CSRC, "ddt_engine_priv.h"), os.path.join(CSRC, "ddt_internal.h")]
    build_if_stale(out, deps, ["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-Wall", *san, "-I" + MOCK, "-I" + CSRC, comm_source,
                               os.path.join(MOCK, "mock_engine.cpp")])
    L = C.CDLL(out)
    L.ddt_create.argtypes, L.ddt_destroy.argtypes, L.ddt_destroy.restype = [C.POINTER(vp), i32], [vp], None
    L.ddt_load_model_shard.argtypes = [vp, C.POINTER(ddt.Params), vp, sz, vp, sz, u32, u32]
    L.ddt_load_model_multiclass.argtypes = [vp, C.POINTER(ddt.Params), vp, sz, vp, sz, u32, i32, u32, u32]
    L.ddt_comm_get_unique_id.argtypes = [vp]
    L.ddt_comm_create.argtypes, L.ddt_comm_destroy.argtypes, L.ddt_comm_destroy.restype = [C.POINTER(vp), vp, i32, i32, vp], [vp], None
    L.ddt_comm_set_option.argtypes = [vp, C.c_char_p, C.c_int64]
    L.ddt_comm_last_error.argtypes, L.ddt_comm_last_error.restype = [vp], C.c_char_p
    L.ddt_score_sharded_device.argtypes = [vp, vp, sz, vp, i32, vp]
    L.ddt_score_rowsharded_device.argtypes = [vp, vp, sz, vp, vp]
    L.ddt_classify_sharded_device.argtypes = [vp, vp, sz, vp, vp, i32, vp]
    L.ddt_comm_score.argtypes = [vp, vp, sz, vp, i32]
    L.ddt_comm_create_hybrid.argtypes = [C.POINTER(vp), vp, i32, i32, i32, vp]
    L.ddt_score_hybrid_device.argtypes = [vp, vp, sz, vp, i32, i32, vp]
    L.ddt_classify_hybrid_device.argtypes = [vp, vp, sz, vp, vp, i32, i32, vp]
    L.ddt_comm_abort.argtypes = [vp]
    L.ddt_hybrid_rows.argtypes = [sz, i32, i32, C.POINTER(sz), C.POINTER(sz)]
    L.ddt_group_create_hybrid.argtypes = [C.POINTER(vp), i32, vp, i32]
    L.ddt_group_create.argtypes, L.ddt_group_destroy.argtypes, L.ddt_group_destroy.restype = [C.POINTER(vp), i32, vp], [vp], None
    L.ddt_group_load_model.argtypes = [vp, C.POINTER(ddt.Params), vp, sz, vp, sz]
    L.ddt_group_load_model_multiclass.argtypes = [vp, C.POINTER(ddt.Params), vp, sz, vp, sz, u32, i32]
    L.ddt_group_score.argtypes = [vp, vp, sz, vp, i32]
    L.ddt_group_classify.argtypes = [vp, vp, sz, vp, vp, i32]
    L.ddt_group_last_error.argtypes, L.ddt_group_last_error.restype = [vp], C.c_char_p
    L.hipSetDevice.argtypes = [i32]
    L.hipStreamCreateWithFlags.argtypes, L.hipStreamSynchronize.argtypes, L.hipStreamDestroy.argtypes = [C.POINTER(vp), C.c_uint], [vp], [vp]
    L.mock_reset.argtypes, L.mock_reset.restype = [i32, u64, i32], None
    L.mock_executed.restype = L.mock_h2d_bytes.restype = u64
    L.mock_costs.argtypes, L.mock_costs.restype = [C.c_double, C.c_double, C.c_double], None
    L.mock_makespan.restype = C.c_double
    return L


@pytest.fixture(scope="module")
def mock():
    return _build("libddt_comm_mock.so", os.path.join(CSRC, "ddt_comm.cpp"))


F, W = 6, 8   # 6 features -> two tuple lines


def _tuples(n):
    x = np.zeros((n, W), np.uint32)
    x[:, 0] = np.arange(n)                       # the stand-in engine scores a tuple by its row id
    x[:, 1:] = 0xDEAD0000
    return x


def _partial(shard, cls, rows):
    return (((rows.astype(np.int64) * 7 + shard * 13 + cls * 101) % 1000) - 500).astype(np.float32)


def _expected(G, n, K=1):
    rows = np.arange(n)
    out = np.zeros((K, n), np.float32)
    for k in range(K):
        for g in range(G):
            out[k] += _partial(g, k, rows)
    return out


def _params(T=64):
    return ddt.make_params(T, 4, F)


def _run_ranks(G, body):
    """body(rank, barrier, shared) on G threads; any exception fails the test."""
    barrier, shared, errors = threading.Barrier(G), {}, []

    def wrap(r):
        try:
            body(r, barrier, shared)
        except BaseException as ex:  # noqa: BLE001
            errors.append((r, repr(ex)))
            barrier.abort()

    th = [threading.Thread(target=wrap, args=(r,)) for r in range(G)]
    for t in th:
        t.start()
    for t in th:
        t.join(120)
    assert not errors, errors
    assert not any(t.is_alive() for t in th), "a rank is stuck"


class Rank:
    """One rank of a process-per-GPU style job: engine + communicator + the caller's stream."""

    def __init__(self, L, r, G, barrier, shared, classes=1, stream_first=True, whole_model=False, tree_ranks=0):
        self.L, self.r, self.G = L, r, G
        assert L.hipSetDevice(r) == 0
        self.s = vp()
        if stream_first:
            assert L.hipStreamCreateWithFlags(C.byref(self.s), 1) == 0
        self.e = vp()
        assert L.ddt_create(C.byref(self.e), r) == 0
        p = _params()
        shard = (0, 1) if whole_model else (r % tree_ranks, tree_ranks) if tree_ranks else (r, G)
        if classes > 1:
            assert L.ddt_load_model_multiclass(self.e, C.byref(p), None, 0, None, 0, classes, 1, *shard) == 0
        else:
            assert L.ddt_load_model_shard(self.e, C.byref(p), None, 0, None, 0, *shard) == 0
        if r == 0:
            shared["id"] = C.create_string_buffer(128)
            assert L.ddt_comm_get_unique_id(shared["id"]) == 0
        barrier.wait()
        self.c = vp()
        if tree_ranks:   # hybrid: row groups of tree_ranks consecutive ranks (ncclCommSplit behind the C-ABI)
            assert L.ddt_comm_create_hybrid(C.byref(self.c), self.e, r, G, tree_ranks, shared["id"]) == 0
        else:
            assert L.ddt_comm_create(C.byref(self.c), self.e, r, G, shared["id"]) == 0
        if not stream_first:
            assert L.hipStreamCreateWithFlags(C.byref(self.s), 1) == 0

    def opt(self, key, value):
        assert self.L.ddt_comm_set_option(self.c, key.encode(), value) == 0

    def sync(self):
        assert self.L.hipStreamSynchronize(self.s) == 0

    def close(self, barrier):
        barrier.wait()                           # nobody tears down while a peer still needs the collective
        self.L.ddt_comm_destroy(self.c)
        self.L.ddt_destroy(self.e)
        self.L.hipStreamDestroy(self.s)


SCHEDULES = [(0, 0), (1, 0), (2, 11), (2, 12), (2, 13)]


@pytest.mark.parametrize("policy,seed", SCHEDULES)
@pytest.mark.parametrize("G,n,chunk", [(1, 5000, 1024), (1, 9000, 12_500_000), (2, 5003, 1024), (3, 1000, 12_500_000), (4, 4097, 700), (8, 3001, 1000),
                                       (8, 1, 5)])
def test_tree_sharded_scores(mock, G, n, chunk, policy, seed):
    mock.mock_reset(policy, seed, 8)
    x, want = _tuples(n), _expected(G, n)[0]

    def body(r, barrier, shared):
        k = Rank(mock, r, G, barrier, shared, stream_first=(seed % 2 == 0))
        k.opt("chunk_rows", chunk)
        k.opt("taper_min_rows", 16)              # the tapered tail (default with peers) at test sizes
        outs = []
        for combine in (0, 1, 1, 0):             # back to back, no host synchronisation in between: workspace slots are reused
            out = np.full(n, np.nan, np.float32)
            assert mock.ddt_score_sharded_device(k.c, x.ctypes.data, n, out.ctypes.data, combine, k.s) == 0, mock.ddt_comm_last_error(k.c)
            outs.append(out)
        barrier.wait()                           # all ranks have enqueued their calls: from here on the schedule decides the order
        k.sync()
        for out in outs:
            assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (r, np.flatnonzero(out != want)[:5])
        k.opt("taper_tail", 0)
        k.opt("comm_stream_priority", 1)          # the comm stream is replaced (after it has drained) by a high-priority one
        out = np.full(n, np.nan, np.float32)
        assert mock.ddt_score_sharded_device(k.c, x.ctypes.data, n, out.ctypes.data, 0, k.s) == 0
        k.sync()
        assert np.array_equal(out.view(np.uint32), want.view(np.uint32))
        k.close(barrier)

    _run_ranks(G, body)
    assert mock.mock_errors() == 0 and mock.mock_executed() > 0


@pytest.mark.parametrize("policy,seed", SCHEDULES)
@pytest.mark.parametrize("G,n,chunk,K", [(1, 2500, 700, 3), (2, 2500, 700, 3), (4, 1029, 12_500_000, 10), (8, 777, 100, 2)])
def test_tree_sharded_classes(mock, G, n, chunk, K, policy, seed):
    mock.mock_reset(policy, seed, 8)
    x, want = _tuples(n), _expected(G, n, K)
    want_labels = np.argmax(want, axis=0).astype(np.int32)      # first maximum = lowest index wins ties

    def body(r, barrier, shared):
        k = Rank(mock, r, G, barrier, shared, classes=K, stream_first=(seed % 2 == 1))
        k.opt("chunk_rows", chunk)
        k.opt("taper_min_rows", 16)
        res = []
        for combine in (0, 1, 0):
            cs, lab = np.full((K, n), np.nan, np.float32), np.full(n, -1, np.int32)
            assert mock.ddt_classify_sharded_device(k.c, x.ctypes.data, n, cs.ctypes.data, lab.ctypes.data, combine, k.s) == 0
            res.append((cs, lab))
        k.sync()
        for cs, lab in res:
            assert np.array_equal(cs.view(np.uint32), want.view(np.uint32)) and np.array_equal(lab, want_labels), r
        k.close(barrier)

    _run_ranks(G, body)
    assert mock.mock_errors() == 0


@pytest.mark.parametrize("policy,seed", SCHEDULES[:3])
@pytest.mark.parametrize("G,n", [(1, 5000), (2, 1000), (3, 1000), (8, 5), (8, 4096), (5, 4099)])
def test_row_sharded_replicas(mock, G, n, policy, seed):
    mock.mock_reset(policy, seed, 8)
    x, want = _tuples(n), _partial(0, 0, np.arange(n))          # every rank holds the whole model (shard 0 of 1)

    def body(r, barrier, shared):
        k = Rank(mock, r, G, barrier, shared, whole_model=True, stream_first=(seed % 2 == 0))
        outs = []
        for chunk in (12_500_000, 300, 64):       # one step; a few; many (every rank sends its step to every peer, in place)
            k.opt("chunk_rows", chunk)
            out = np.full(n, np.nan, np.float32)
            assert mock.ddt_score_rowsharded_device(k.c, x.ctypes.data, n, out.ctypes.data, k.s) == 0
            outs.append(out)
        k.sync()
        for out in outs:
            assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), r
        k.close(barrier)

    _run_ranks(G, body)
    assert mock.mock_errors() == 0


@pytest.mark.parametrize("policy,seed", SCHEDULES[:3])
def test_host_buffer_form(mock, policy, seed):
    G, n = 4, 6007
    mock.mock_reset(policy, seed, 8)
    x, want = _tuples(n), _expected(G, n)[0]

    def body(r, barrier, shared):
        k = Rank(mock, r, G, barrier, shared)
        k.opt("host_rows", 2500)                 # three super-chunks, the last one ragged
        k.opt("chunk_rows", 900)
        k.opt("taper_min_rows", 16)
        for bcast in (1, 0):                     # tuples cross "PCIe" once and travel on between the devices / every rank copies all
            k.opt("tuple_broadcast", bcast)
            barrier.wait()
            before = mock.mock_h2d_bytes()
            barrier.wait()
            for combine in (0, 1):
                out = np.full(n, np.nan, np.float32)
                assert mock.ddt_comm_score(k.c, x.ctypes.data, n, out.ctypes.data, combine) == 0
                assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (r, bcast)
            barrier.wait()
            if r == 0:
                traffic.append((mock.mock_h2d_bytes() - before) / (2 * n * W * 4))
            barrier.wait()
        k.close(barrier)

    traffic = []
    _run_ranks(G, body)
    assert mock.mock_errors() == 0
    assert traffic == [1.0, float(G)]            # host tuples: once with the broadcast, G times without


@pytest.mark.parametrize("policy,seed", SCHEDULES[:3])
@pytest.mark.parametrize("G", [2, 8])
def test_single_process_group(mock, G, policy, seed):
    """ddt_group_*: one process, one worker thread per device (ncclCommInitAll)."""
    mock.mock_reset(policy, seed, 8)
    n, K = 9001, 4
    x = _tuples(n)
    g = vp()
    assert mock.ddt_group_create(C.byref(g), G, None) == 0
    p = _params()
    assert mock.ddt_group_load_model(g, C.byref(p), x.ctypes.data, 1 << 20, x.ctypes.data, 1 << 20) == 0
    for combine in (0, 1):
        out = np.full(n, np.nan, np.float32)
        assert mock.ddt_group_score(g, x.ctypes.data, n, out.ctypes.data, combine) == 0, mock.ddt_group_last_error(g)
        assert np.array_equal(out.view(np.uint32), _expected(G, n)[0].view(np.uint32))
    assert mock.ddt_group_load_model_multiclass(g, C.byref(p), x.ctypes.data, 1 << 20, x.ctypes.data, 1 << 20, K, 1) == 0
    want = _expected(G, n, K)
    lab, cs = np.full(n, -1, np.int32), np.full((K, n), np.nan, np.float32)
    assert mock.ddt_group_classify(g, x.ctypes.data, n, lab.ctypes.data, cs.ctypes.data, 0) == 0
    assert np.array_equal(cs.view(np.uint32), want.view(np.uint32)) and np.array_equal(lab, np.argmax(want, axis=0))
    mock.ddt_group_load_model_replicated.argtypes = [vp, C.POINTER(ddt.Params), vp, sz, vp, sz]
    mock.ddt_group_score_rows.argtypes = [vp, vp, sz, vp]
    out = np.full(n, np.nan, np.float32)
    assert mock.ddt_group_score_rows(g, x.ctypes.data, n, out.ctypes.data) < 0            # tree shards (or classes) loaded: refused
    assert mock.ddt_group_load_model_replicated(g, C.byref(p), x.ctypes.data, 1 << 20, x.ctypes.data, 1 << 20) == 0
    assert mock.ddt_group_score_rows(g, x.ctypes.data, n, out.ctypes.data) == 0
    assert np.array_equal(out.view(np.uint32), _partial(0, 0, np.arange(n)).view(np.uint32))
    mock.ddt_group_destroy(g)
    assert mock.mock_errors() == 0


def test_teardown_in_either_order(mock):
    """ddt_comm_destroy must not read the engine: a caller (or a garbage collector) may destroy the engine first."""
    mock.mock_reset(0, 0, 8)

    def body(r, barrier, shared):
        k = Rank(mock, r, 2, barrier, shared)
        x, out = _tuples(100), np.zeros(100, np.float32)
        assert mock.ddt_score_sharded_device(k.c, x.ctypes.data, 100, out.ctypes.data, 1, k.s) == 0
        k.sync()
        barrier.wait()
        mock.ddt_destroy(k.e)                     # engine first ...
        mock.ddt_comm_destroy(k.c)                # ... then the communicator (run under DDT_MOCK_SANITIZE to see a stale read)
        mock.hipStreamDestroy(k.s)

    _run_ranks(2, body)


def _makespan(mock, G, n, chunk, call, taper, cost_row=1.0, cost_float=0.25):
    """ASAP timeline of one call on every rank: scoring costs `cost_row` per row, a collective `cost_float` per float a rank
    moves (mock_runtime.cpp "Timeline")."""
    mock.mock_reset(0, 0, 8)
    x = _tuples(n)
    spans = []

    def body(r, barrier, shared):
        k = Rank(mock, r, G, barrier, shared, whole_model=(call == "rows"))
        k.opt("chunk_rows", chunk)
        k.opt("taper_tail", taper)
        k.opt("taper_min_rows", 16)
        barrier.wait()
        if r == 0:
            mock.mock_costs(cost_row, cost_float, 0.0)           # clocks start here
        barrier.wait()
        out = np.full(n, np.nan, np.float32)
        if call == "rows":
            assert mock.ddt_score_rowsharded_device(k.c, x.ctypes.data, n, out.ctypes.data, k.s) == 0
        else:
            assert mock.ddt_score_sharded_device(k.c, x.ctypes.data, n, out.ctypes.data, 1 if call == "chain" else 0, k.s) == 0
        k.sync()
        barrier.wait()
        if r == 0:
            spans.append(mock.mock_makespan())
            mock.mock_costs(0.0, 0.0, 0.0)
        k.close(barrier)

    _run_ranks(G, body)
    return spans[0]


def test_only_the_last_collective_is_exposed():
    """The overlap the pipeline is built for, read off the model's clock: with a chunk's collective cheaper than the next chunk's
    scoring, a tree-sharded call takes the scoring of all rows plus ONE collective -- the last piece's, a quarter chunk with the
    tapered tail -- and a row-sharded call the scoring of n / G rows plus the last step's messages."""
    mock = _build("libddt_comm_mock.so", os.path.join(CSRC, "ddt_comm.cpp"))
    G, n, chunk, cf = 8, 80_000, 10_000, 0.25
    plain = _makespan(mock, G, n, chunk, "allreduce", 0)
    assert plain == pytest.approx(n * 1.0 + chunk * cf)                       # 8 chunks scored back to back + the 8th all-reduce
    L = ddt.lib()
    k = L.ddt_comm_chunk_schedule(n, chunk, 1, 16, None, 0)
    lens = np.zeros(k, np.uint64)
    L.ddt_comm_chunk_schedule(n, chunk, 1, 16, lens.ctypes.data, k)
    tapered = _makespan(mock, G, n, chunk, "allreduce", 1)
    assert tapered == pytest.approx(n * 1.0 + int(lens[-1]) * cf) and int(lens[-1]) <= chunk // 4 + 1024
    assert tapered < plain
    chain = _makespan(mock, G, n, chunk, "chain", 1)                           # all-to-all + all-gather: two collectives per piece
    assert n * 1.0 < chain <= n * 1.0 + 2 * int(lens[-1]) * cf + 1
    rows = _makespan(mock, G, n, chunk, "rows", 0)
    step = -(-chunk // G)
    step = -(-step // 1024) * 1024 if step >= 1024 else step
    last = (n // G) - ((n // G - 1) // step) * step                            # rows of the last step of a rank
    assert rows == pytest.approx(n / G * 1.0 + last * cf)                       # one message per link at once
    assert rows < tapered / 4                                                   # replicas: an eighth of the rows per rank


REMOVED_WAITS = {
    # the collective of a chunk no longer waits for the chunk's scoring launch (in-place all-reduce path)
    "scored_inplace": ("        CHIP(c, hipStreamWaitEvent(c->cs, c->ev_scored[b], 0));\n        CNCCL(c, ncclAllReduce(dst + lo, dst + lo, m, ncclFloat, ncclSum, c->comm, c->cs));",
                       "        CNCCL(c, ncclAllReduce(dst + lo, dst + lo, m, ncclFloat, ncclSum, c->comm, c->cs));", 0),
    # ... the same on the staged path (chain combine / classes)
    "scored_staged": ("        CHIP(c, hipEventRecord(c->ev_scored[b], s));\n        CHIP(c, hipStreamWaitEvent(c->cs, c->ev_scored[b], 0));\n        const float* res;",
                      "        const float* res;", 1),
    # the scoring launch of chunk k+2 no longer waits until slot b has been consumed by chunk k's collective
    "slot_free": ("        if (c->slot_used[b]) CHIP(c, hipStreamWaitEvent(s, c->ev_free[b], 0));  // slot b still feeds chunk k-2's collective\n", "", 1),
    # the caller's stream no longer waits for the last collective
    "done": ("  CHIP(c, hipStreamWaitEvent(s, c->ev_done, 0));  // results are ready in stream order on the caller's stream\n", "", 1),
    # row-sharded job: a step is handed to the peers before it has been scored / the caller does not wait for the peers' rows
    "rows_scored": ("    CHIP(c, hipStreamWaitEvent(c->cs, c->ev_scored[b], 0));\n    GroupGuard grp;\n    CNCCL(c, grp.start());", "    GroupGuard grp;\n    CNCCL(c, grp.start());", -1),
    "rows_done": ("    CHIP(c, hipStreamWaitEvent(s, c->ev_done, 0));  // every peer's rows have landed before the caller's stream moves on\n", "", -1),
}


@pytest.mark.skipif(bool(os.environ.get("DDT_MOCK_SANITIZE")), reason="the broken builds race on purpose")
@pytest.mark.parametrize("which,hybrid", [(w, False) for w in sorted(REMOVED_WAITS)] + [("scored_inplace", True), ("scored_staged", True), ("done", True)])
def test_the_model_catches_a_missing_dependency(which, hybrid):
    """Remove ONE wait of the event protocol from the source: some schedule must then give wrong scores -- the proof that
    the schedules above would have exposed such a hole in the shipped pipeline.  hybrid: the same holes seen through a 2 x 2 hybrid job
    (the pieces a row group hands to the other one come from the same streams)."""
    needle, repl, combine = REMOVED_WAITS[which]
    src = open(os.path.join(CSRC, "ddt_comm.cpp")).read()
    assert src.count(needle) == 1, which
    broken, so = os.path.join(MOCK, f"_broken_{which}_{int(hybrid)}.cpp"), f"libddt_comm_mock_broken_{which}_{int(hybrid)}.so"
    open(broken, "w").write(src.replace(needle, repl))
    try:
        bad = _build(so, broken)
        G, n = (4, 5000) if hybrid else (2, 3000)
        x, want = _tuples(n), (_expected(2, n)[0] if combine >= 0 else _partial(0, 0, np.arange(n)))
        wrong = 0
        for policy, seed in SCHEDULES:
            bad.mock_reset(policy, seed, 8)
            seen = []

            def body(r, barrier, shared):
                k = Rank(bad, r, G, barrier, shared, stream_first=True, whole_model=combine < 0, tree_ranks=2 if hybrid else 0)
                k.opt("chunk_rows", 300)
                outs = []
                for _ in range(2):
                    out = np.full(n, np.nan, np.float32)
                    if hybrid:
                        assert bad.ddt_score_hybrid_device(k.c, x.ctypes.data, n, out.ctypes.data, combine, 1, k.s) == 0
                    elif combine < 0:
                        assert bad.ddt_score_rowsharded_device(k.c, x.ctypes.data, n, out.ctypes.data, k.s) == 0
                    else:
                        assert bad.ddt_score_sharded_device(k.c, x.ctypes.data, n, out.ctypes.data, combine, k.s) == 0
                    outs.append(out)
                barrier.wait()                    # every rank has enqueued everything: the schedule, not thread timing, decides the order
                k.sync()
                seen.append(all(np.array_equal(o.view(np.uint32), want.view(np.uint32)) for o in outs))
                k.close(barrier)

            _run_ranks(G, body)
            wrong += not all(seen)
        assert wrong > 0, f"no schedule noticed the missing wait ({which}, hybrid {hybrid})"
    finally:
        for f in (broken, os.path.join(MOCK, so)):
            if os.path.exists(f):
                os.remove(f)


def _rows(mock, n, Gr, rg):
    lo, hi = sz(), sz()
    assert mock.ddt_hybrid_rows(n, Gr, rg, C.byref(lo), C.byref(hi)) == 0
    return lo.value, hi.value


@pytest.mark.parametrize("policy,seed", SCHEDULES)
@pytest.mark.parametrize("Gt,Gr,n,chunk", [(2, 2, 6000, 700), (2, 4, 9001, 1024), (4, 2, 5003, 300), (1, 4, 4100, 500), (8, 1, 3001, 1000), (2, 4, 1500, 12_500_000),
                                           (2, 2, 3, 5)])
def test_hybrid_job(mock, Gt, Gr, n, chunk, policy, seed):
    """ddt_comm_create_hybrid + ddt_score_hybrid_device: Gr row groups of Gt consecutive ranks; a row group is a tree-sharded job on its
    slice of the rows (communicator split off the world communicator), the finished pieces travel to the other row groups while the
    next piece is scored.  gather: every rank ends up with all rows; no gather: a rank writes its row group's rows only.  Ragged row
    counts (a last group that is shorter, or empty), both combines, back-to-back calls on the reused workspace slots."""
    mock.mock_reset(policy, seed, 8)
    x, want = _tuples(n), _expected(Gt, n)[0]          # every tuple: the sum of the Gt shards' partials (the groups hold the same Gt shards)

    def body(r, barrier, shared):
        k = Rank(mock, r, Gt * Gr, barrier, shared, stream_first=(seed % 2 == 0), tree_ranks=Gt)
        k.opt("chunk_rows", chunk)
        k.opt("taper_min_rows", 16)
        lo, hi = _rows(mock, n, Gr, r // Gt)
        outs = []
        for combine, gather in ((0, 1), (1, 1), (0, 0), (1, 0), (0, 1)):
            out = np.full(n, np.nan, np.float32)
            assert mock.ddt_score_hybrid_device(k.c, x.ctypes.data, n, out.ctypes.data, combine, gather, k.s) == 0, mock.ddt_comm_last_error(k.c)
            outs.append((out, gather))
        barrier.wait()                               # all ranks have enqueued their calls: from here on the schedule decides the order
        k.sync()
        for out, gather in outs:
            if gather:
                assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), (r, np.flatnonzero(out != want)[:5])
            else:
                assert np.array_equal(out[lo:hi], want[lo:hi]) and np.isnan(out[:lo]).all() and np.isnan(out[hi:]).all(), r
        # the other jobs' calls refuse a hybrid communicator
        out = np.zeros(n, np.float32)
        assert mock.ddt_score_sharded_device(k.c, x.ctypes.data, n, out.ctypes.data, 0, k.s) == -4
        assert mock.ddt_score_rowsharded_device(k.c, x.ctypes.data, n, out.ctypes.data, k.s) == -4
        k.close(barrier)

    _run_ranks(Gt * Gr, body)
    assert mock.mock_errors() == 0 and mock.mock_executed() > 0


@pytest.mark.parametrize("policy,seed", SCHEDULES[:3])
@pytest.mark.parametrize("Gt,Gr,n,chunk,K", [(2, 2, 2500, 700, 3), (2, 4, 3333, 400, 10), (4, 2, 1029, 12_500_000, 2)])
def test_hybrid_classes(mock, Gt, Gr, n, chunk, K, policy, seed):
    mock.mock_reset(policy, seed, 8)
    x, want = _tuples(n), _expected(Gt, n, K)
    want_labels = np.argmax(want, axis=0).astype(np.int32)

    def body(r, barrier, shared):
        k = Rank(mock, r, Gt * Gr, barrier, shared, classes=K, stream_first=(seed % 2 == 1), tree_ranks=Gt)
        k.opt("chunk_rows", chunk)
        k.opt("taper_min_rows", 16)
        lo, hi = _rows(mock, n, Gr, r // Gt)
        res = []
        for combine, gather in ((0, 1), (1, 0), (1, 1)):
            cs, lab = np.full((K, n), np.nan, np.float32), np.full(n, -1, np.int32)
            assert mock.ddt_classify_hybrid_device(k.c, x.ctypes.data, n, cs.ctypes.data, lab.ctypes.data, combine, gather, k.s) == 0
            res.append((cs, lab, gather))
        k.sync()
        for cs, lab, gather in res:
            if gather:
                assert np.array_equal(cs.view(np.uint32), want.view(np.uint32)) and np.array_equal(lab, want_labels), r
            else:
                assert np.array_equal(cs[:, lo:hi], want[:, lo:hi]) and np.array_equal(lab[lo:hi], want_labels[lo:hi]), r
                assert np.isnan(cs[:, :lo]).all() and np.isnan(cs[:, hi:]).all() and (lab[:lo] == -1).all() and (lab[hi:] == -1).all()
        k.close(barrier)

    _run_ranks(Gt * Gr, body)
    assert mock.mock_errors() == 0


@pytest.mark.parametrize("policy,seed", SCHEDULES[:3])
def test_hybrid_host_buffer_form_moves_the_tuples_once(mock, policy, seed):
    """ddt_comm_score on a hybrid communicator: a rank copies 1 / n_ranks of every super-chunk over "PCIe" (its share of its row group's
    slice) and the row group hands the rows on; every rank gets all scores."""
    Gt, Gr, n = 2, 2, 6007
    mock.mock_reset(policy, seed, 8)
    x, want = _tuples(n), _expected(Gt, n)[0]
    traffic = []

    def body(r, barrier, shared):
        k = Rank(mock, r, Gt * Gr, barrier, shared, tree_ranks=Gt)
        k.opt("host_rows", 2500)
        k.opt("chunk_rows", 300)
        k.opt("taper_min_rows", 16)
        barrier.wait()
        before = mock.mock_h2d_bytes()
        barrier.wait()
        for combine in (0, 1):
            out = np.full(n, np.nan, np.float32)
            assert mock.ddt_comm_score(k.c, x.ctypes.data, n, out.ctypes.data, combine) == 0, mock.ddt_comm_last_error(k.c)
            assert np.array_equal(out.view(np.uint32), want.view(np.uint32)), r
        barrier.wait()
        if r == 0:
            traffic.append((mock.mock_h2d_bytes() - before) / (2 * n * W * 4))
        k.close(barrier)

    _run_ranks(Gt * Gr, body)
    assert mock.mock_errors() == 0 and traffic == [1.0]


@pytest.mark.parametrize("policy,seed", SCHEDULES[:3])
@pytest.mark.parametrize("G,Gt", [(4, 2), (8, 2), (8, 4)])
def test_single_process_hybrid_group(mock, G, Gt, policy, seed):
    """ddt_group_create_hybrid: device i holds tree shard i % Gt of Gt; ddt_group_score / ddt_group_classify run the hybrid job."""
    mock.mock_reset(policy, seed, 8)
    n, K = 9001, 4
    x = _tuples(n)
    g = vp()
    assert mock.ddt_group_create_hybrid(C.byref(g), G, None, 3) == -1        # 3 does not divide the device count
    assert mock.ddt_group_create_hybrid(C.byref(g), G, None, Gt) == 0
    p = _params()
    assert mock.ddt_group_load_model(g, C.byref(p), x.ctypes.data, 1 << 20, x.ctypes.data, 1 << 20) == 0
    for combine in (0, 1):
        out = np.full(n, np.nan, np.float32)
        assert mock.ddt_group_score(g, x.ctypes.data, n, out.ctypes.data, combine) == 0, mock.ddt_group_last_error(g)
        assert np.array_equal(out.view(np.uint32), _expected(Gt, n)[0].view(np.uint32))
    assert mock.ddt_group_load_model_multiclass(g, C.byref(p), x.ctypes.data, 1 << 20, x.ctypes.data, 1 << 20, K, 1) == 0
    want = _expected(Gt, n, K)
    lab, cs = np.full(n, -1, np.int32), np.full((K, n), np.nan, np.float32)
    assert mock.ddt_group_classify(g, x.ctypes.data, n, lab.ctypes.data, cs.ctypes.data, 0) == 0
    assert np.array_equal(cs.view(np.uint32), want.view(np.uint32)) and np.array_equal(lab, np.argmax(want, axis=0))
    mock.ddt_group_destroy(g)
    assert mock.mock_errors() == 0


def test_a_rank_that_fails_before_its_collective(mock):
    """2 x 2 hybrid job; rank 3's call fails before it has enqueued anything (here: a NULL tuple pointer -> DDT_EINVAL).  Its peers
    have queued collectives that can never complete: rank 2's all-reduce needs rank 3, ranks 0 / 1 wait for row group 1's pieces.  The
    launcher tells them; ddt_comm_abort frees their streams, the communicator refuses further calls, teardown works in any order."""
    mock.mock_reset(0, 0, 8)
    Gt, G, n = 2, 4, 4000
    x = _tuples(n)
    failed = threading.Event()

    def body(r, barrier, shared):
        k = Rank(mock, r, G, barrier, shared, tree_ranks=Gt)
        k.opt("chunk_rows", 500)
        out = np.full(n, np.nan, np.float32)
        barrier.wait()
        if r == 3:
            assert mock.ddt_score_hybrid_device(k.c, None, n, out.ctypes.data, 0, 1, k.s) == -1      # refused before anything is enqueued
            failed.set()
        else:
            assert mock.ddt_score_hybrid_device(k.c, x.ctypes.data, n, out.ctypes.data, 0, 1, k.s) == 0
            assert failed.wait(30)
            assert mock.ddt_comm_abort(k.c) == 0                          # told by the launcher that a peer is gone
            k.sync()                                                      # returns: the queued collectives ended without running
            assert mock.ddt_score_hybrid_device(k.c, x.ctypes.data, n, out.ctypes.data, 0, 1, k.s) == -4     # dead communicator
            assert mock.ddt_comm_abort(k.c) == 0                          # idempotent
        barrier.wait()
        mock.ddt_comm_destroy(k.c)
        mock.ddt_destroy(k.e)
        mock.hipStreamDestroy(k.s)

    _run_ranks(G, body)


def _fuzz_job(mock, seed):
    rng = np.random.default_rng(seed)
    mock.mock_reset(int(rng.integers(0, 3)), int(rng.integers(0, 1000)), 8)
    G = int(rng.choice([1, 2, 3, 4, 5, 8]))
    K = int(rng.choice([1, 1, 1, 2, 5]))
    rows_mode = K == 1 and rng.random() < 0.25
    plan = []
    for _ in range(int(rng.integers(1, 5))):
        plan.append(dict(n=int(rng.choice([1, 5, 700, 1024, 4097, 9001])), chunk=int(rng.choice([5, 64, 700, 1024, 12_500_000])), taper=int(rng.choice([-1, 0, 1])),
                         tmin=int(rng.choice([1, 16, 1024])), combine=int(rng.integers(0, 2)), host=bool(rng.random() < 0.3 and K == 1 and not rows_mode), prio=int(rng.integers(0, 2)),
                         host_rows=int(rng.choice([300, 2500, 8 << 20])), bcast=int(rng.choice([-1, 0, 1])), sync=bool(rng.random() < 0.5)))
    stream_first = bool(rng.integers(0, 2))
    def body(r, barrier, shared):
        k = Rank(mock, r, G, barrier, shared, classes=K, stream_first=stream_first, whole_model=rows_mode)
        keep = []
        for st in plan:
            n = st["n"]
            x = _tuples(n)
            k.opt("chunk_rows", st["chunk"]); k.opt("taper_tail", st["taper"]); k.opt("taper_min_rows", st["tmin"]); k.opt("comm_stream_priority", st["prio"])
            k.opt("host_rows", st["host_rows"]); k.opt("tuple_broadcast", st["bcast"])
            if K > 1:
                cs, lab = np.full((K, n), np.nan, np.float32), np.full(n, -1, np.int32)
                assert mock.ddt_classify_sharded_device(k.c, x.ctypes.data, n, cs.ctypes.data, lab.ctypes.data, st["combine"], k.s) == 0, mock.ddt_comm_last_error(k.c)
                keep.append((x, "cls", cs, lab, n))
            elif rows_mode:
                out = np.full(n, np.nan, np.float32)
                assert mock.ddt_score_rowsharded_device(k.c, x.ctypes.data, n, out.ctypes.data, k.s) == 0, mock.ddt_comm_last_error(k.c)
                keep.append((x, "rows", out, None, n))
            elif st["host"]:
                out = np.full(n, np.nan, np.float32)
                assert mock.ddt_comm_score(k.c, x.ctypes.data, n, out.ctypes.data, st["combine"]) == 0, mock.ddt_comm_last_error(k.c)
                keep.append((x, "tree", out, None, n))
            else:
                out = np.full(n, np.nan, np.float32)
                assert mock.ddt_score_sharded_device(k.c, x.ctypes.data, n, out.ctypes.data, st["combine"], k.s) == 0, mock.ddt_comm_last_error(k.c)
                keep.append((x, "tree", out, None, n))
            if st["sync"]:
                k.sync()
        barrier.wait()
        k.sync()
        for x, kind, a, b, n in keep:
            if kind == "cls":
                want = _expected(G, n, K)
                assert np.array_equal(a.view(np.uint32), want.view(np.uint32)) and np.array_equal(b, np.argmax(want, axis=0).astype(np.int32)), (seed, r, kind, n)
            elif kind == "rows":
                assert np.array_equal(a.view(np.uint32), _partial(0, 0, np.arange(n)).view(np.uint32)), (seed, r, kind, n)
            else:
                assert np.array_equal(a.view(np.uint32), _expected(G, n)[0].view(np.uint32)), (seed, r, kind, n)
        k.close(barrier)
    _run_ranks(G, body)
    assert mock.mock_errors() == 0, seed


@pytest.mark.parametrize("seed", [1, 2, 3, 5, 8, 13, 21, 34])
def test_random_jobs_options_and_call_sequences(mock, seed):
    """A trimmed copy of a fuzz loop run once over 400 seeds without a failure: 1-8 ranks, scalar / class / replica jobs, resident and
    host calls of awkward sizes back to back, chunk sizes from 5 rows up, tapered and untapered tails, both combines, stream priority,
    tuple broadcast on and off, random schedules."""
    _fuzz_job(mock, seed)
