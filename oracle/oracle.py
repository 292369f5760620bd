"""ctypes binding of oracle/liboracle.so -- the CPU ORACLE (test infrastructure, NOT product code).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module, and only
as the checker / the reported CPU baseline.  Parity: arithmetic, compare rule, traversal datapath, schedule and the
programming side pinned against vectors evaluated / executed from the reference's RTL source, the handshake / FIFO control
UNPINNED (the reference ships no tests or golden vectors and cannot be simulated as a whole here); see oracle/ddt_oracle.h.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")

SUM_REF_FLOPOCO, SUM_REF_NATIVE, SUM_F64_SEQ = 0, 1, 2


class Params(C.Structure):
    """Mirror of orc_params (== the reference's CSR 204/205 fields, EngineCSR.sv:218-233)."""

    _fields_ = [
        ("num_trees", C.c_uint32),
        ("num_levels", C.c_uint32),
        ("num_features", C.c_uint32),
        ("missing_bits", C.c_uint32),
        ("weights_lines_per_tree", C.c_uint32),
        ("findex_lines_per_tree", C.c_uint32),
        ("cmp_mode", C.c_uint32),
        ("clusters_per_tuple", C.c_uint32),
    ]


def build(force: bool = False) -> str:
    src = [os.path.join(_HERE, f) for f in ("ddt_oracle.c", "ddt_oracle.h", "Makefile")]
    stale = (not os.path.exists(_LIB_PATH)) or any(
        os.path.getmtime(s) > os.path.getmtime(_LIB_PATH) for s in src
    )
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s", "liboracle.so"])
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u32, u64, vp, sz = C.c_uint32, C.c_uint64, C.c_void_p, C.c_size_t
        PP = C.POINTER(Params)
        L.orc_fp34_wrap.restype, L.orc_fp34_wrap.argtypes = u64, [u32]
        L.orc_fp34_unwrap.restype, L.orc_fp34_unwrap.argtypes = u32, [u64]
        L.orc_fp34_add.restype, L.orc_fp34_add.argtypes = u64, [u64, u64]
        L.orc_fpadd_bits.restype, L.orc_fpadd_bits.argtypes = u32, [u32, u32]
        L.orc_go_right.restype, L.orc_go_right.argtypes = u32, [u32, u32, u32, u32, u32]
        L.orc_tree8.restype, L.orc_tree8.argtypes = u32, [vp]
        L.orc_aggregate.restype, L.orc_aggregate.argtypes = u32, [vp, u32]
        L.orc_fpadd_bits_batch.restype, L.orc_fpadd_bits_batch.argtypes = None, [vp, vp, vp, sz]
        L.orc_traverse.restype, L.orc_traverse.argtypes = u32, [PP, vp, vp, vp, u32]
        L.orc_leaves.restype, L.orc_leaves.argtypes = None, [PP, vp, vp, vp, vp]
        L.orc_leaves_fast.restype, L.orc_leaves_fast.argtypes = None, [PP, vp, vp, vp, vp]
        L.orc_reduce_device.restype, L.orc_reduce_device.argtypes = u32, [vp, u32, u32, C.c_int]
        L.orc_score.restype = C.c_int
        L.orc_score.argtypes = [PP, vp, sz, vp, sz, vp, sz, vp, vp, C.c_int, C.c_int, C.c_int]
        L.orc_score_fast.restype = C.c_int
        L.orc_score_fast.argtypes = [PP, vp, sz, vp, sz, vp, sz, vp, C.c_int, C.c_int]
        L.orc_score_fast_ex.restype = C.c_int
        L.orc_score_fast_ex.argtypes = [PP, vp, sz, vp, sz, vp, sz, vp, C.c_int, C.c_int, u32, C.c_int, vp, vp, vp, vp, C.c_int]
        L.orc_fast_add_selftest.restype, L.orc_fast_add_selftest.argtypes = u64, [u64, u64]
        L.orc_score_shard.restype = C.c_int
        L.orc_score_shard.argtypes = [PP, vp, sz, vp, sz, vp, sz, u32, u32, vp, C.c_int, C.c_int]
        L.orc_classify.restype = C.c_int
        L.orc_classify.argtypes = [PP, vp, sz, vp, sz, vp, sz, u32, C.c_int, C.c_int, C.c_int, vp, vp]
        for n in ("orc_weights_lines_per_tree", "orc_findex_lines_per_tree", "orc_tuple_lines"):
            getattr(L, n).restype, getattr(L, n).argtypes = u32, [u32]
        L.orc_pack_model.restype, L.orc_pack_model.argtypes = None, [u32, u32, vp, vp, vp, vp, vp, vp]
        L.orc_splitmix64.restype, L.orc_splitmix64.argtypes = u64, [u64]
        L.orc_gen_tuples.restype, L.orc_gen_tuples.argtypes = None, [u64, sz, u32, C.c_int, u32, vp]
        L.orc_gen_model.restype, L.orc_gen_model.argtypes = None, [u32, u32, u32, C.c_int, vp, vp]
        L.orc_hw_threads.restype, L.orc_hw_threads.argtypes = C.c_int, []
        L.orc_sparse_check.restype, L.orc_sparse_check.argtypes = C.c_int, [PP, vp, sz, vp]
        L.orc_traverse_sparse.restype, L.orc_traverse_sparse.argtypes = u32, [PP, vp, vp, vp, u32]
        L.orc_sparse_mean_depth.restype, L.orc_sparse_mean_depth.argtypes = C.c_double, [PP, vp, vp, vp, sz]
        L.orc_score_sparse.restype = C.c_int
        L.orc_score_sparse.argtypes = [PP, vp, sz, vp, vp, sz, vp, vp, C.c_int, C.c_int, C.c_int]
        L.orc_classify_sparse.restype = C.c_int
        L.orc_classify_sparse.argtypes = [PP, vp, sz, vp, vp, sz, u32, C.c_int, C.c_int, C.c_int, vp, vp]
        L.orc_score_sparse_fast.restype = C.c_int
        L.orc_score_sparse_fast.argtypes = [PP, vp, sz, vp, vp, sz, vp, C.c_int, C.c_int]
        L.orc_sparse_from_perfect.restype, L.orc_sparse_from_perfect.argtypes = None, [PP, vp, vp, vp, vp]
        L.orc_sparse_to_perfect.restype, L.orc_sparse_to_perfect.argtypes = C.c_int, [PP, vp, vp, vp, vp]
        L.orc_gen_sparse_model.restype = sz
        L.orc_gen_sparse_model.argtypes = [u32, u32, u32, u32, u32, C.c_int, vp, sz, vp]
        _lib = L
    return _lib


def _p(a: np.ndarray):
    return a.ctypes.data_as(C.c_void_p)


def wlpt(D: int) -> int:
    return ((1 << (D + 1)) - 1 + 3) // 4


def flpt(D: int) -> int:
    return ((1 << D) - 1 + 7) // 8


def tuple_lines(F: int) -> int:
    return (F + 3) // 4


def default_clusters(T: int) -> int:
    """Smallest C in {1,2,4,8} whose 8*16*C tree slots hold T trees (16 trees/PU, DTPU.sv:74)."""
    for c in (1, 2, 4, 8):
        if T <= 128 * c:
            return c
    return 8


def make_params(T, D, F, missing_bits=0x7FC00000, cmp_mode=0, clusters=None) -> Params:
    return Params(T, D, F, missing_bits, wlpt(D), flpt(D), cmp_mode,
                  default_clusters(T) if clusters is None else clusters)


class Model:
    """A model in the reference wire format: weights lines (u32) + feature-index lines (u16)."""

    def __init__(self, params: Params, wlines: np.ndarray, flines: np.ndarray):
        self.params = params
        self.wlines = np.ascontiguousarray(wlines, dtype=np.uint32).reshape(-1)
        self.flines = np.ascontiguousarray(flines, dtype=np.uint16).reshape(-1)

    @property
    def n_wlines(self) -> int:
        return self.wlines.size // 4

    @property
    def n_flines(self) -> int:
        return self.flines.size // 8


def pack_model(thr: np.ndarray, fidx: np.ndarray, miss_right: np.ndarray, leaves: np.ndarray,
               F: int, **kw) -> Model:
    """thr [T, 2^D-1] fp32 (or uint32 bit patterns), fidx [T, 2^D-1], miss_right [T, 2^D-1], leaves [T, 2^D]."""
    T, nint = fidx.shape
    D = int(np.log2(nint + 1))
    assert (1 << D) - 1 == nint and leaves.shape == (T, 1 << D)
    tb = np.ascontiguousarray(thr).view(np.uint32) if thr.dtype == np.float32 else np.ascontiguousarray(thr, np.uint32)
    lb = np.ascontiguousarray(leaves).view(np.uint32) if leaves.dtype == np.float32 else np.ascontiguousarray(leaves, np.uint32)
    fi = np.ascontiguousarray(fidx, np.uint16)
    mr = np.ascontiguousarray(miss_right, np.uint8)
    w = np.zeros(T * wlpt(D) * 4, np.uint32)
    f = np.zeros(T * flpt(D) * 8, np.uint16)
    lib().orc_pack_model(T, D, _p(tb), _p(fi), _p(mr), _p(lb), _p(w), _p(f))
    return Model(make_params(T, D, F, **kw), w, f)


def gen_model(T: int, D: int, F: int, dist: int = 0, **kw) -> Model:
    w = np.zeros(T * wlpt(D) * 4, np.uint32)
    f = np.zeros(T * flpt(D) * 8, np.uint16)
    lib().orc_gen_model(T, D, F, dist, _p(w), _p(f))
    return Model(make_params(T, D, F, **kw), w, f)


def gen_tuples(row0: int, n: int, F: int, dist: int = 0, missing_bits: int = 0x7FC00000) -> np.ndarray:
    """-> uint32 [n, ceil(F/4)*4] tuple lines (fp32 bit patterns)."""
    out = np.zeros((n, tuple_lines(F) * 4), np.uint32)
    lib().orc_gen_tuples(row0, n, F, dist, missing_bits, _p(out))
    return out


def tuples_from_float(x: np.ndarray) -> np.ndarray:
    """fp32 [n, F] -> uint32 tuple lines [n, ceil(F/4)*4] (zero padded)."""
    n, F = x.shape
    out = np.zeros((n, tuple_lines(F) * 4), np.uint32)
    out[:, :F] = np.ascontiguousarray(x, np.float32).view(np.uint32)
    return out


def score(m: Model, tuples: np.ndarray, sum_mode: int = SUM_REF_FLOPOCO, n_devices: int = 1,
          nthreads: int = 0, want_gold: bool = False):
    t = np.ascontiguousarray(tuples, np.uint32)
    n = t.shape[0]
    out = np.zeros(n, np.float32)
    gold = np.zeros(n, np.float64) if want_gold else None
    rc = lib().orc_score(C.byref(m.params), _p(m.wlines), m.n_wlines, _p(m.flines), m.n_flines, _p(t), n,
                         _p(out), _p(gold) if want_gold else None, sum_mode, n_devices, nthreads)
    if rc:
        raise ValueError(f"orc_score rc={rc}")
    return (out, gold) if want_gold else out


def score_fast(m: Model, tuples: np.ndarray, sum_mode: int = SUM_REF_NATIVE, nthreads: int = 0, n_devices: int = 1, want_gold: bool = False):
    """The CPU-baseline form of score(): identical bits, cache-blocked (oracle/ddt_oracle.c sections 8 and 8b).
    want_gold: -> (scores, gold, gold_abs): per row the fp64 sum of its leaves (score()'s gold) and of their magnitudes."""
    t = np.ascontiguousarray(tuples, np.uint32)
    out = np.zeros(t.shape[0], np.float32)
    if m.params.cmp_mode == 0 and sum_mode != SUM_F64_SEQ and (sum_mode == SUM_REF_FLOPOCO or n_devices > 1 or want_gold):
        gold = np.zeros(t.shape[0], np.float64) if want_gold else None
        gabs = np.zeros(t.shape[0], np.float64) if want_gold else None
        rc = lib().orc_score_fast_ex(C.byref(m.params), _p(m.wlines), m.n_wlines, _p(m.flines), m.n_flines, _p(t), t.shape[0],
                                     _p(out), sum_mode, n_devices, 0, 1, None, None, _p(gold) if want_gold else None,
                                     _p(gabs) if want_gold else None, nthreads)
        if rc:
            raise ValueError(f"orc_score_fast_ex rc={rc}")
        return (out, gold, gabs) if want_gold else out
    if n_devices > 1 or want_gold:
        if want_gold:
            raise ValueError("score_fast(want_gold): cmp_mode 0 and a reference-order sum only")
        return score(m, t, sum_mode=sum_mode, n_devices=n_devices, nthreads=nthreads)
    rc = lib().orc_score_fast(C.byref(m.params), _p(m.wlines), m.n_wlines, _p(m.flines), m.n_flines, _p(t), t.shape[0],
                              _p(out), sum_mode, nthreads)
    if rc:
        raise ValueError(f"orc_score_fast rc={rc}")
    return out


def score_shard(m: Model, tuples: np.ndarray, tree_begin: int, tree_end: int,
                sum_mode: int = SUM_REF_FLOPOCO, nthreads: int = 0) -> np.ndarray:
    t = np.ascontiguousarray(tuples, np.uint32)
    out = np.zeros(t.shape[0], np.float32)
    rc = lib().orc_score_shard(C.byref(m.params), _p(m.wlines), m.n_wlines, _p(m.flines), m.n_flines, _p(t),
                               t.shape[0], tree_begin, tree_end, _p(out), sum_mode, nthreads)
    if rc:
        raise ValueError(f"orc_score_shard rc={rc}")
    return out


def classify(m: Model, tuples: np.ndarray, num_classes: int, interleaved: bool = True,
             sum_mode: int = SUM_REF_FLOPOCO, n_devices: int = 1):
    """-> (labels int32 [n], class_scores fp32 [K, n])"""
    t = np.ascontiguousarray(tuples, np.uint32)
    n = t.shape[0]
    labels = np.zeros(n, np.int32)
    cs = np.zeros((num_classes, n), np.float32)
    rc = lib().orc_classify(C.byref(m.params), _p(m.wlines), m.n_wlines, _p(m.flines), m.n_flines, _p(t), n,
                            num_classes, int(interleaved), sum_mode, n_devices, _p(labels), _p(cs))
    if rc:
        raise ValueError(f"orc_classify rc={rc}")
    return labels, cs


def classify_fast(m: Model, tuples: np.ndarray, num_classes: int, interleaved: bool = True,
                  sum_mode: int = SUM_REF_NATIVE, n_devices: int = 1, nthreads: int = 0, want_gold: bool = False):
    """classify() in the cache-blocked form (identical bits; oracle/ddt_oracle.c section 8b).
    want_gold: -> (labels, class_scores, gold [K, n], gold_abs [K, n])"""
    if m.params.cmp_mode != 0 or sum_mode == SUM_F64_SEQ:
        if want_gold:
            raise ValueError("classify_fast(want_gold): cmp_mode 0 and a reference-order sum only")
        return classify(m, tuples, num_classes, interleaved, sum_mode, n_devices)
    t = np.ascontiguousarray(tuples, np.uint32)
    n = t.shape[0]
    labels = np.zeros(n, np.int32)
    cs = np.zeros((num_classes, n), np.float32)
    gold = np.zeros((num_classes, n), np.float64) if want_gold else None
    gabs = np.zeros((num_classes, n), np.float64) if want_gold else None
    rc = lib().orc_score_fast_ex(C.byref(m.params), _p(m.wlines), m.n_wlines, _p(m.flines), m.n_flines, _p(t), n,
                                 None, sum_mode, n_devices, num_classes, int(interleaved), _p(labels), _p(cs),
                                 _p(gold) if want_gold else None, _p(gabs) if want_gold else None, nthreads)
    if rc:
        raise ValueError(f"orc_score_fast_ex rc={rc}")
    return (labels, cs, gold, gabs) if want_gold else (labels, cs)


def fast_add_selftest(seed: int, n: int) -> int:
    return int(lib().orc_fast_add_selftest(seed, n))


def leaves(m: Model, tuple_row: np.ndarray) -> np.ndarray:
    t = np.ascontiguousarray(tuple_row, np.uint32)
    out = np.zeros(m.params.num_trees, np.uint32)
    lib().orc_leaves(C.byref(m.params), _p(m.wlines), _p(m.flines), _p(t), _p(out))
    return out


def leaves_fast(m: Model, tuple_row: np.ndarray) -> np.ndarray:
    t = np.ascontiguousarray(tuple_row, np.uint32)
    out = np.zeros(m.params.num_trees, np.uint32)
    lib().orc_leaves_fast(C.byref(m.params), _p(m.wlines), _p(m.flines), _p(t), _p(out))
    return out


def fpadd_bits(a: int, b: int) -> int:
    return lib().orc_fpadd_bits(a & 0xFFFFFFFF, b & 0xFFFFFFFF)


def fpadd_bits_batch(a: np.ndarray, b: np.ndarray) -> np.ndarray:
    a = np.ascontiguousarray(a, np.uint32)
    b = np.ascontiguousarray(b, np.uint32)
    out = np.zeros_like(a)
    lib().orc_fpadd_bits_batch(_p(a), _p(b), _p(out), a.size)
    return out


def hw_threads() -> int:
    return lib().orc_hw_threads()


# ---- sparse (explicit-children) model stream: this repository's extension, see ddt_oracle.h ----------------------
class SparseModel:
    """node_lines uint32 [n_lines, 4] (one 128-bit line per internal node), tree_first_line uint64 [T + 1]."""

    def __init__(self, params: Params, node_lines: np.ndarray, tree_first_line: np.ndarray):
        self.params = params
        self.node_lines = np.ascontiguousarray(node_lines, dtype=np.uint32).reshape(-1, 4)
        self.first = np.ascontiguousarray(tree_first_line, dtype=np.uint64).reshape(-1)
        assert self.first.size == params.num_trees + 1

    @property
    def n_lines(self) -> int:
        return self.node_lines.shape[0]

    def check(self) -> int:
        return lib().orc_sparse_check(C.byref(self.params), _p(self.node_lines), self.n_lines, _p(self.first))


def make_sparse_params(T, max_depth, F, missing_bits=0x7FC00000, cmp_mode=0, clusters=None) -> Params:
    return Params(T, max_depth, F, missing_bits, 0, 0, cmp_mode, default_clusters(T) if clusters is None else clusters)


def gen_sparse_model(T: int, max_depth: int, F: int, full_levels: int, split_permille: int, dist: int = 0, **kw) -> SparseModel:
    first = np.zeros(T + 1, np.uint64)
    n = lib().orc_gen_sparse_model(T, max_depth, F, full_levels, split_permille, dist, None, 0, _p(first))
    lines = np.zeros((n, 4), np.uint32)
    lib().orc_gen_sparse_model(T, max_depth, F, full_levels, split_permille, dist, _p(lines), n, _p(first))
    return SparseModel(make_sparse_params(T, max_depth, F, **kw), lines, first)


def sparse_from_perfect(m: Model) -> SparseModel:
    T, D = m.params.num_trees, m.params.num_levels
    lines = np.zeros((T * ((1 << D) - 1), 4), np.uint32)
    first = np.zeros(T + 1, np.uint64)
    lib().orc_sparse_from_perfect(C.byref(m.params), _p(m.wlines), _p(m.flines), _p(lines), _p(first))
    q = m.params
    return SparseModel(Params(T, D, q.num_features, q.missing_bits, 0, 0, q.cmp_mode, q.clusters_per_tuple), lines, first)


def sparse_to_perfect(s: SparseModel, D: int | None = None) -> Model:
    """pad_to_perfect"""
    q = s.params
    D = q.num_levels if D is None else D
    pp = Params(q.num_trees, D, q.num_features, q.missing_bits, wlpt(D), flpt(D), q.cmp_mode, q.clusters_per_tuple)
    w = np.zeros(q.num_trees * wlpt(D) * 4, np.uint32)
    f = np.zeros(q.num_trees * flpt(D) * 8, np.uint16)
    rc = lib().orc_sparse_to_perfect(C.byref(pp), _p(s.node_lines), _p(s.first), _p(w), _p(f))
    if rc:
        raise ValueError(f"orc_sparse_to_perfect rc={rc}")
    return Model(pp, w, f)


def score_sparse(s: SparseModel, tuples: np.ndarray, sum_mode: int = SUM_REF_FLOPOCO, n_devices: int = 1,
                 nthreads: int = 0, want_gold: bool = False):
    t = np.ascontiguousarray(tuples, np.uint32)
    n = t.shape[0]
    out = np.zeros(n, np.float32)
    gold = np.zeros(n, np.float64) if want_gold else None
    rc = lib().orc_score_sparse(C.byref(s.params), _p(s.node_lines), s.n_lines, _p(s.first), _p(t), n, _p(out),
                                _p(gold) if want_gold else None, sum_mode, n_devices, nthreads)
    if rc:
        raise ValueError(f"orc_score_sparse rc={rc}")
    return (out, gold) if want_gold else out


def score_sparse_fast(s: SparseModel, tuples: np.ndarray, sum_mode: int = SUM_REF_NATIVE, nthreads: int = 0) -> np.ndarray:
    """The CPU-baseline form of score_sparse(): identical bits, one tree at a time over a block of rows."""
    t = np.ascontiguousarray(tuples, np.uint32)
    out = np.zeros(t.shape[0], np.float32)
    rc = lib().orc_score_sparse_fast(C.byref(s.params), _p(s.node_lines), s.n_lines, _p(s.first), _p(t), t.shape[0], _p(out),
                                     sum_mode, nthreads)
    if rc:
        raise ValueError(f"orc_score_sparse_fast rc={rc}")
    return out


def classify_sparse(s: SparseModel, tuples: np.ndarray, num_classes: int, interleaved: bool = True,
                    sum_mode: int = SUM_REF_FLOPOCO, n_devices: int = 1):
    """-> (labels int32 [n], class_scores fp32 [K, n])"""
    t = np.ascontiguousarray(tuples, np.uint32)
    n = t.shape[0]
    labels = np.zeros(n, np.int32)
    cs = np.zeros((num_classes, n), np.float32)
    rc = lib().orc_classify_sparse(C.byref(s.params), _p(s.node_lines), s.n_lines, _p(s.first), _p(t), n, num_classes,
                                   int(interleaved), sum_mode, n_devices, _p(labels), _p(cs))
    if rc:
        raise ValueError(f"orc_classify_sparse rc={rc}")
    return labels, cs


def traverse_sparse(s: SparseModel, tuple_row: np.ndarray, tree: int) -> int:
    t = np.ascontiguousarray(tuple_row, np.uint32)
    return lib().orc_traverse_sparse(C.byref(s.params), _p(s.node_lines), _p(s.first), _p(t), tree)


def sparse_mean_depth(s: SparseModel, tuples: np.ndarray) -> float:
    """mean node visits per (tuple, tree) on these tuples"""
    t = np.ascontiguousarray(tuples, np.uint32)
    return float(lib().orc_sparse_mean_depth(C.byref(s.params), _p(s.node_lines), _p(s.first), _p(t), t.shape[0]))
