"""Host-side mirror of the engine interface above the C-ABI (include/ddt.h).

The reference has no host software; what a host program must do is fixed by its hardware contract:
write the run parameters (CSR 201-207, rtl/DTEngine/EngineCSR.sv:190-248), stream the model
(weights lines then feature-index lines, rtl/DTEngine/PCIeReceiver.sv:136-139,230-275), stream tuples,
read back result lines (rtl/DTEngine/ResultsCombiner.sv:136-160).  `Engine` is that sequence as an
object: Engine(device) -> load_model(params, weights_lines, findex_lines) -> score(tuple_lines).
PyTorch is used only for device memory and streams.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import Info, Params, Stats

MISSING_DEFAULT = 0x7FC00000


class DDTError(RuntimeError):
    def __init__(self, code: int, detail: str = ""):
        self.code = code
        msg = _lib.lib().ddt_strerror(code).decode()
        super().__init__(f"ddt error {code} ({msg}){': ' + detail if detail else ''}")


def weights_lines_per_tree(D: int) -> int:
    return ((1 << (D + 1)) - 1 + 3) // 4


def findex_lines_per_tree(D: int) -> int:
    return ((1 << D) - 1 + 7) // 8


def tuple_words(F: int) -> int:
    return (F + 3) // 4 * 4


def shard_bounds(T: int, G: int):
    """Contiguous shards of ceil(T/G) trees in stream order: the split ddt_load_model_shard applies (PCIeReceiver.sv:241-264)."""
    per = (T + G - 1) // G
    return [(min(g * per, T), min((g + 1) * per, T)) for g in range(G)]


def default_clusters(T: int) -> int:
    """Smallest C in {1,2,4,8} whose 8 PUs x 16 trees x C slots hold T trees (DTPU.sv:74, Core.sv:291-316)."""
    for c in (1, 2, 4, 8):
        if T <= 128 * c:
            return c
    return 8


def make_params(T: int, D: int, F: int, missing_bits: int = MISSING_DEFAULT, cmp_mode: int = 0,
                clusters: int | None = None, sum_mode: int = 0) -> Params:
    p = Params()
    p.num_trees, p.num_levels, p.num_features, p.missing_bits = T, D, F, missing_bits
    p.weights_lines_per_tree, p.findex_lines_per_tree = weights_lines_per_tree(D), findex_lines_per_tree(D)
    p.cmp_mode, p.sum_mode = cmp_mode, sum_mode
    p.clusters_per_tuple = default_clusters(T) if clusters is None else clusters
    return p


def synth_model(T: int, D: int, F: int, dist: int = 0):
    """Deterministic synthetic model of SURVEY.md 8(d) in the reference wire format -> (wlines u32, flines u16)."""
    w = np.zeros(T * weights_lines_per_tree(D) * 4, np.uint32)
    f = np.zeros(T * findex_lines_per_tree(D) * 8, np.uint16)
    rc = _lib.lib().ddt_synth_model(T, D, F, dist, w.ctypes.data, f.ctypes.data)
    if rc:
        raise DDTError(rc)
    return w, f


def make_sparse_params(T: int, max_depth: int, F: int, missing_bits: int = MISSING_DEFAULT, cmp_mode: int = 0,
                       clusters: int | None = None, sum_mode: int = 0) -> Params:
    """Parameters of a SPARSE (explicit-children) model: num_levels is the depth bound, no lines-per-tree fields."""
    p = Params()
    p.num_trees, p.num_levels, p.num_features, p.missing_bits = T, max_depth, F, missing_bits
    p.cmp_mode, p.sum_mode = cmp_mode, sum_mode
    p.clusters_per_tuple = default_clusters(T) if clusters is None else clusters
    return p


def synth_sparse_model(T: int, max_depth: int, F: int, full_levels: int, split_permille: int, dist: int = 0):
    """Deterministic random-forest-like sparse model (include/ddt.h ddt_synth_sparse_model)
    -> (node_lines uint32 [n, 4], tree_first_line uint64 [T + 1])."""
    L = _lib.lib()
    first = np.zeros(T + 1, np.uint64)
    n = L.ddt_synth_sparse_model(T, max_depth, F, full_levels, split_permille, dist, None, 0, first.ctypes.data)
    if n < 0:
        raise DDTError(int(n))
    lines = np.zeros((n, 4), np.uint32)
    L.ddt_synth_sparse_model(T, max_depth, F, full_levels, split_permille, dist, lines.ctypes.data, n, first.ctypes.data)
    return lines, first


def synth_tuples_host(row0: int, n: int, F: int, dist: int = 0, missing_bits: int = MISSING_DEFAULT) -> np.ndarray:
    out = np.zeros((n, tuple_words(F)), np.uint32)
    rc = _lib.lib().ddt_synth_tuples_host(out.ctypes.data, row0, n, F, dist, missing_bits)
    if rc:
        raise DDTError(rc)
    return out


class Engine:
    """One engine == one GPU (the analogue of one FPGA running DTInference)."""

    def __init__(self, device: int = 0):
        self._L = _lib.lib()
        h = C.c_void_p()
        rc = self._L.ddt_create(C.byref(h), device)
        if rc:
            raise DDTError(rc, "ddt_create")
        self._h = h
        self.device = device
        self.params = None
        self.num_classes = 1

    def close(self):
        if getattr(self, "_h", None):
            self._L.ddt_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc:
            raise DDTError(rc, self._L.ddt_last_error(self._h).decode())

    # ---- model ------------------------------------------------------------------------------------
    def load_model(self, params: Params, wlines: np.ndarray, flines: np.ndarray, shard_index: int = 0,
                   shard_count: int = 1):
        w = np.ascontiguousarray(wlines).view(np.uint32).reshape(-1)
        f = np.ascontiguousarray(flines).view(np.uint16).reshape(-1)
        self._check(self._L.ddt_load_model_shard(self._h, C.byref(params), w.ctypes.data, w.size // 4,
                                                 f.ctypes.data, f.size // 8, shard_index, shard_count))
        self.params = params
        self.num_classes = 1
        return self

    def load_model_sparse(self, params: Params, node_lines: np.ndarray, tree_first_line: np.ndarray,
                          shard_index: int = 0, shard_count: int = 1, num_classes: int = 1, interleaved: bool = True):
        """Sparse (explicit-children) forest: one 128-bit line per internal node (include/ddt.h ddt_load_model_sparse);
        num_classes > 1: one-vs-all classes in the stream (ddt_load_model_sparse_multiclass), scored with classify*()."""
        nl = np.ascontiguousarray(node_lines).view(np.uint32).reshape(-1, 4)
        first = np.ascontiguousarray(tree_first_line, dtype=np.uint64).reshape(-1)
        if first.size != params.num_trees + 1:
            raise ValueError("tree_first_line must hold num_trees + 1 entries")
        self._check(self._L.ddt_load_model_sparse_multiclass(self._h, C.byref(params), nl.ctypes.data, nl.shape[0], first.ctypes.data,
                                                             num_classes, int(interleaved), shard_index, shard_count))
        self.params = params
        self.num_classes = num_classes
        return self

    def load_model_multiclass(self, params: Params, wlines: np.ndarray, flines: np.ndarray, num_classes: int,
                              interleaved: bool = True, shard_index: int = 0, shard_count: int = 1):
        """One-vs-all ensemble of `num_classes` classes (BASELINE config 5; an extension of the reference)."""
        w = np.ascontiguousarray(wlines).view(np.uint32).reshape(-1)
        f = np.ascontiguousarray(flines).view(np.uint16).reshape(-1)
        self._check(self._L.ddt_load_model_multiclass(self._h, C.byref(params), w.ctypes.data, w.size // 4,
                                                      f.ctypes.data, f.size // 8, num_classes, int(interleaved),
                                                      shard_index, shard_count))
        self.params = params
        self.num_classes = num_classes
        return self

    def classify(self, tuple_lines: np.ndarray, want_scores: bool = False):
        """Host buffers -> int32 labels [n] (and fp32 class scores [K, n])."""
        t = np.ascontiguousarray(tuple_lines).view(np.uint32).reshape(-1, tuple_words(self.params.num_features))
        n = t.shape[0]
        labels = np.empty(n, np.int32)
        cs = np.empty((self.num_classes, n), np.float32) if want_scores else None
        self._check(self._L.ddt_classify(self._h, t.ctypes.data, n, labels.ctypes.data,
                                         cs.ctypes.data if want_scores else None))
        return (labels, cs) if want_scores else labels

    def classify_device(self, d_tuples, class_scores=None, labels=None, want_labels: bool = True, stream=None):
        """torch CUDA tuple lines -> (labels int32 [n] or None, class scores fp32 [K, n]); asynchronous."""
        import torch

        W = tuple_words(self.params.num_features)
        n = d_tuples.numel() // W
        if class_scores is None:
            class_scores = torch.empty((self.num_classes, n), dtype=torch.float32, device=d_tuples.device)
        if labels is None and want_labels:
            labels = torch.empty(n, dtype=torch.int32, device=d_tuples.device)
        s = torch.cuda.current_stream(d_tuples.device) if stream is None else stream
        self._check(self._L.ddt_classify_device(self._h, d_tuples.data_ptr(), n, class_scores.data_ptr(),
                                                labels.data_ptr() if labels is not None else None, s.cuda_stream))
        return labels, class_scores

    def argmax_device(self, class_scores, labels=None, stream=None):
        import torch

        K, n = class_scores.shape
        if labels is None:
            labels = torch.empty(n, dtype=torch.int32, device=class_scores.device)
        s = torch.cuda.current_stream(class_scores.device) if stream is None else stream
        self._check(self._L.ddt_argmax_device(self._h, class_scores.data_ptr(), K, n, labels.data_ptr(), s.cuda_stream))
        return labels

    def set_option(self, key: str, value: int):
        self._check(self._L.ddt_set_option(self._h, key.encode(), int(value)))

    # ---- scoring ----------------------------------------------------------------------------------
    def score(self, tuple_lines: np.ndarray, out: np.ndarray | None = None) -> np.ndarray:
        """Host buffers through the pinned, pipelined feeder (the PCIe tuple stream); buffers pinned with host_register skip the staging."""
        t = np.ascontiguousarray(tuple_lines).view(np.uint32)
        if self.params is None:  # let the library report DDT_ESTATE
            self._check(self._L.ddt_score(self._h, t.ctypes.data, 1, t.ctypes.data))
        W = tuple_words(self.params.num_features)
        t = t.reshape(-1, W)
        if out is None:
            out = np.empty(t.shape[0], np.float32)
        assert out.dtype == np.float32 and out.flags.c_contiguous and out.size >= t.shape[0]
        self._check(self._L.ddt_score(self._h, t.ctypes.data, t.shape[0], out.ctypes.data))
        return out

    def host_register(self, a: np.ndarray):
        """Pin a host array for this engine's device (ddt_host_register): ddt_score then DMAs it directly."""
        assert a.flags.c_contiguous
        self._check(self._L.ddt_host_register(self._h, a.ctypes.data, a.nbytes))

    def host_unregister(self, a: np.ndarray):
        self._check(self._L.ddt_host_unregister(self._h, a.ctypes.data))

    def score_device(self, d_tuples, out=None, stream=None):
        """torch CUDA tensor of tuple lines ([n, W] int32/uint32/float32) -> torch float32 [n], asynchronous."""
        import torch

        W = tuple_words(self.params.num_features)
        assert d_tuples.is_cuda and d_tuples.is_contiguous() and d_tuples.element_size() == 4
        n = d_tuples.numel() // W
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=d_tuples.device)
        assert out.is_cuda and out.is_contiguous() and out.dtype == torch.float32 and out.numel() >= n
        s = torch.cuda.current_stream(d_tuples.device) if stream is None else stream
        self._check(self._L.ddt_score_device(self._h, d_tuples.data_ptr(), n, out.data_ptr(), s.cuda_stream))
        return out

    def chain_sum_device(self, parts, out=None, stream=None):
        """parts: torch float32 [G, n] -> out[n] = (((p0+p1)+p2)+...) in the reference's chain order."""
        import torch

        assert parts.is_cuda and parts.is_contiguous() and parts.dtype == torch.float32 and parts.dim() == 2
        G, n = parts.shape
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=parts.device)
        s = torch.cuda.current_stream(parts.device) if stream is None else stream
        self._check(self._L.ddt_chain_sum_device(self._h, parts.data_ptr(), G, n, out.data_ptr(), s.cuda_stream))
        return out

    def synth_tuples_device(self, row0: int, n: int, F: int, dist: int = 0, missing_bits: int = MISSING_DEFAULT,
                            out=None, stream=None):
        import torch

        W = tuple_words(F)
        if out is None:
            out = torch.empty((n, W), dtype=torch.int32, device=f"cuda:{self.device}")
        s = torch.cuda.current_stream(out.device) if stream is None else stream
        self._check(self._L.ddt_synth_tuples_device(self._h, out.data_ptr(), row0, n, F, dist, missing_bits,
                                                    s.cuda_stream))
        return out

    # ---- introspection ----------------------------------------------------------------------------
    def info(self) -> Info:
        i = Info()
        self._check(self._L.ddt_get_info(self._h, C.byref(i)))
        return i

    def stats(self) -> Stats:
        s = Stats()
        self._check(self._L.ddt_get_stats(self._h, C.byref(s)))
        return s


def variant_names():
    L = _lib.lib()
    out = []
    for v in range(L.ddt_num_variants()):
        b = C.create_string_buffer(64)
        L.ddt_variant_name(v, b, 64)
        out.append(b.value.decode())
    return out


COMBINE_ALLREDUCE, COMBINE_CHAIN = 0, 1
COMM_ID_BYTES = 128


def hybrid_rows(n: int, row_groups: int, row_group: int):
    """rows [lo, hi) of a row group of a hybrid job (ddt_hybrid_rows)"""
    lo, hi = C.c_size_t(), C.c_size_t()
    rc = _lib.lib().ddt_hybrid_rows(n, row_groups, row_group, C.byref(lo), C.byref(hi))
    if rc:
        raise DDTError(rc, "ddt_hybrid_rows")
    return lo.value, hi.value


def comm_unique_id() -> bytes:
    """ncclGetUniqueId through the C-ABI: rank 0 makes it, the launcher hands it to every rank."""
    buf = C.create_string_buffer(COMM_ID_BYTES)
    rc = _lib.lib().ddt_comm_get_unique_id(buf)
    if rc:
        raise DDTError(rc, "ddt_comm_get_unique_id")
    return buf.raw


class Comm:
    """One rank of a multi-GPU job: an RCCL communicator bound to an Engine (include/ddt.h ddt_comm_*).  The sharded
    calls are collective: every rank calls them with the same arguments."""

    def __init__(self, engine: Engine, rank: int, n_ranks: int, unique_id: bytes, tree_ranks: int = 0):
        """tree_ranks = 0: a plain communicator (tree-sharded over all ranks, or row-sharded replicas).  tree_ranks = Gt >= 1: the hybrid
        layout (ddt_comm_create_hybrid) -- row groups of Gt consecutive ranks, the engine must hold tree shard rank % Gt of Gt."""
        self._L, self.engine, self.rank, self.n_ranks, self.tree_ranks = _lib.lib(), engine, rank, n_ranks, tree_ranks
        assert len(unique_id) == COMM_ID_BYTES
        h = C.c_void_p()
        if tree_ranks:
            rc = self._L.ddt_comm_create_hybrid(C.byref(h), engine._h, rank, n_ranks, tree_ranks, unique_id)
        else:
            rc = self._L.ddt_comm_create(C.byref(h), engine._h, rank, n_ranks, unique_id)
        if rc:
            raise DDTError(rc, self._L.ddt_last_error(engine._h).decode())
        self._h = h

    def layout(self):
        lay = _lib.CommLayout()
        self._check(self._L.ddt_comm_layout(self._h, C.byref(lay)))
        return lay

    def abort(self):
        """A peer failed: free this rank's queued collectives (ddt_comm_abort); the communicator is dead afterwards."""
        self._check(self._L.ddt_comm_abort(self._h))

    def score_hybrid(self, d_tuples, out=None, combine: int = COMBINE_ALLREDUCE, gather: bool = True, stream=None):
        """Hybrid job: this rank's row group scores its slice of the rows against the group's tree shards; gather: all rows on every rank."""
        import torch

        n, s = self._args(d_tuples, stream)
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=d_tuples.device)
        self._check(self._L.ddt_score_hybrid_device(self._h, d_tuples.data_ptr(), n, out.data_ptr(), combine, 1 if gather else 0, s.cuda_stream))
        return out

    def classify_hybrid(self, d_tuples, combine: int = COMBINE_ALLREDUCE, gather: bool = True, want_labels: bool = True, stream=None,
                        class_scores=None, labels=None):
        import torch

        n, s = self._args(d_tuples, stream)
        cs = class_scores if class_scores is not None else torch.empty((self.engine.num_classes, n), dtype=torch.float32, device=d_tuples.device)
        if labels is None and want_labels:
            labels = torch.empty(n, dtype=torch.int32, device=d_tuples.device)
        want_labels = labels is not None
        self._check(self._L.ddt_classify_hybrid_device(self._h, d_tuples.data_ptr(), n, cs.data_ptr(), labels.data_ptr() if want_labels else None,
                                                       combine, 1 if gather else 0, s.cuda_stream))
        return labels, cs

    def close(self):
        if getattr(self, "_h", None):
            self._L.ddt_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc:
            raise DDTError(rc, self._L.ddt_comm_last_error(self._h).decode())

    def set_option(self, key: str, value: int):
        self._check(self._L.ddt_comm_set_option(self._h, key.encode(), int(value)))

    def _args(self, d_tuples, stream):
        import torch

        W = tuple_words(self.engine.params.num_features)
        assert d_tuples.is_cuda and d_tuples.is_contiguous() and d_tuples.element_size() == 4
        s = torch.cuda.current_stream(d_tuples.device) if stream is None else stream
        return d_tuples.numel() // W, s

    def score_sharded(self, d_tuples, out=None, combine: int = COMBINE_ALLREDUCE, stream=None):
        """Tree-sharded job: tuples replicated on every rank -> full fp32 scores on every rank (asynchronous)."""
        import torch

        n, s = self._args(d_tuples, stream)
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=d_tuples.device)
        self._check(self._L.ddt_score_sharded_device(self._h, d_tuples.data_ptr(), n, out.data_ptr(), combine, s.cuda_stream))
        return out

    def score(self, tuple_lines: np.ndarray, combine: int = COMBINE_ALLREDUCE) -> np.ndarray:
        """Host buffers (ddt_comm_score): with peers the tuples cross PCIe once (1/n per rank) and travel on over xGMI; every rank
        gets the combined scores.  Collective and synchronous."""
        t = np.ascontiguousarray(tuple_lines).view(np.uint32).reshape(-1, tuple_words(self.engine.params.num_features))
        out = np.empty(t.shape[0], np.float32)
        self._check(self._L.ddt_comm_score(self._h, t.ctypes.data, t.shape[0], out.ctypes.data, combine))
        return out

    def score_rowsharded(self, d_tuples, out=None, stream=None):
        """Row-sharded job ("replicas only"): every rank holds the whole ensemble and scores its slice of the rows."""
        import torch

        n, s = self._args(d_tuples, stream)
        if out is None:
            out = torch.empty(n, dtype=torch.float32, device=d_tuples.device)
        self._check(self._L.ddt_score_rowsharded_device(self._h, d_tuples.data_ptr(), n, out.data_ptr(), s.cuda_stream))
        return out

    def classify_sharded(self, d_tuples, combine: int = COMBINE_ALLREDUCE, want_labels: bool = True, stream=None,
                         class_scores=None, labels=None):
        """-> (labels int32 [n] or None, combined class scores fp32 [K, n])"""
        import torch

        n, s = self._args(d_tuples, stream)
        cs = class_scores if class_scores is not None else torch.empty((self.engine.num_classes, n), dtype=torch.float32, device=d_tuples.device)
        if labels is None and want_labels:
            labels = torch.empty(n, dtype=torch.int32, device=d_tuples.device)
        want_labels = labels is not None
        self._check(self._L.ddt_classify_sharded_device(self._h, d_tuples.data_ptr(), n, cs.data_ptr(),
                                                        labels.data_ptr() if want_labels else None, combine, s.cuda_stream))
        return labels, cs


class Group:
    """Single-process multi-GPU job (include/ddt.h ddt_group_*): n engines + one RCCL communicator over them."""

    def __init__(self, device_ids, tree_ranks: int = 0):
        """tree_ranks = Gt >= 1: the hybrid layout (row groups of Gt consecutive devices, ddt_group_create_hybrid)"""
        self._L = _lib.lib()
        ids = (C.c_int * len(device_ids))(*device_ids)
        h = C.c_void_p()
        rc = (self._L.ddt_group_create_hybrid(C.byref(h), len(device_ids), ids, tree_ranks) if tree_ranks
              else self._L.ddt_group_create(C.byref(h), len(device_ids), ids))
        if rc:
            raise DDTError(rc, "ddt_group_create")
        self._h, self.n, self.params = h, len(device_ids), None

    def close(self):
        if getattr(self, "_h", None):
            self._L.ddt_group_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _check(self, rc: int):
        if rc:
            raise DDTError(rc, self._L.ddt_group_last_error(self._h).decode())

    def load_model(self, params: Params, wlines: np.ndarray, flines: np.ndarray):
        w = np.ascontiguousarray(wlines).view(np.uint32).reshape(-1)
        f = np.ascontiguousarray(flines).view(np.uint16).reshape(-1)
        self._check(self._L.ddt_group_load_model(self._h, C.byref(params), w.ctypes.data, w.size // 4, f.ctypes.data, f.size // 8))
        self.params = params
        return self

    def load_model_sparse(self, params: Params, node_lines: np.ndarray, tree_first_line: np.ndarray):
        nl = np.ascontiguousarray(node_lines).view(np.uint32).reshape(-1, 4)
        first = np.ascontiguousarray(tree_first_line, dtype=np.uint64).reshape(-1)
        self._check(self._L.ddt_group_load_model_sparse(self._h, C.byref(params), nl.ctypes.data, nl.shape[0], first.ctypes.data))
        self.params = params
        return self

    def load_model_multiclass(self, params: Params, wlines: np.ndarray, flines: np.ndarray, num_classes: int, interleaved: bool = True):
        w = np.ascontiguousarray(wlines).view(np.uint32).reshape(-1)
        f = np.ascontiguousarray(flines).view(np.uint16).reshape(-1)
        self._check(self._L.ddt_group_load_model_multiclass(self._h, C.byref(params), w.ctypes.data, w.size // 4, f.ctypes.data, f.size // 8,
                                                             num_classes, 1 if interleaved else 0))
        self.params, self.num_classes = params, num_classes
        return self

    def load_model_replicated(self, params: Params, wlines: np.ndarray, flines: np.ndarray):
        """Every device holds the whole ensemble (the reference's tuple-partitioned mode): use with score_rows()."""
        w = np.ascontiguousarray(wlines).view(np.uint32).reshape(-1)
        f = np.ascontiguousarray(flines).view(np.uint16).reshape(-1)
        self._check(self._L.ddt_group_load_model_replicated(self._h, C.byref(params), w.ctypes.data, w.size // 4, f.ctypes.data, f.size // 8))
        self.params = params
        return self

    def score_rows(self, tuple_lines: np.ndarray) -> np.ndarray:
        """Device d scores its share of the rows through its own feeder; no collective."""
        t = np.ascontiguousarray(tuple_lines).view(np.uint32).reshape(-1, tuple_words(self.params.num_features))
        out = np.empty(t.shape[0], np.float32)
        self._check(self._L.ddt_group_score_rows(self._h, t.ctypes.data, t.shape[0], out.ctypes.data))
        return out

    def score(self, tuple_lines: np.ndarray, combine: int = COMBINE_ALLREDUCE) -> np.ndarray:
        t = np.ascontiguousarray(tuple_lines).view(np.uint32).reshape(-1, tuple_words(self.params.num_features))
        out = np.empty(t.shape[0], np.float32)
        self._check(self._L.ddt_group_score(self._h, t.ctypes.data, t.shape[0], out.ctypes.data, combine))
        return out

    def classify(self, tuple_lines: np.ndarray, combine: int = COMBINE_ALLREDUCE, want_scores: bool = True):
        """labels [n] int32 and (optionally) the combined per-class sums [num_classes][n]."""
        t = np.ascontiguousarray(tuple_lines).view(np.uint32).reshape(-1, tuple_words(self.params.num_features))
        labels = np.empty(t.shape[0], np.int32)
        cs = np.empty((self.num_classes, t.shape[0]), np.float32) if want_scores else None
        self._check(self._L.ddt_group_classify(self._h, t.ctypes.data, t.shape[0], labels.ctypes.data, cs.ctypes.data if want_scores else None, combine))
        return labels, cs
