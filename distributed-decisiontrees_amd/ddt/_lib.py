"""Loader for lib/libddt.so (the C-ABI of include/ddt.h).  Fails loudly when the library is missing:
there is no Python/CPU fallback for the scoring path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

_PKG = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB_PATH = os.path.join(_PKG, "lib", "libddt.so")
CSRC = os.path.join(_PKG, "csrc")

# every symbol include/ddt.h declares (tests check the built library exports exactly these)
SYMBOLS = [
    "ddt_create", "ddt_destroy", "ddt_load_model", "ddt_load_model_shard", "ddt_score", "ddt_score_device",
    "ddt_chain_sum_device", "ddt_load_model_multiclass", "ddt_classify_device", "ddt_classify", "ddt_argmax_device",
    "ddt_csr_encode", "ddt_csr_decode", "ddt_get_info", "ddt_get_stats", "ddt_strerror", "ddt_last_error", "ddt_set_option",
    "ddt_num_variants", "ddt_variant_name", "ddt_synth_model", "ddt_synth_tuples_host", "ddt_synth_tuples_device",
    "ddt_load_model_sparse", "ddt_load_model_sparse_multiclass", "ddt_synth_sparse_model", "ddt_csr_encode_ex", "ddt_csr_decode_ex",
    "ddt_comm_get_unique_id", "ddt_comm_create", "ddt_comm_destroy", "ddt_comm_last_error", "ddt_comm_set_option",
    "ddt_score_sharded_device", "ddt_score_rowsharded_device", "ddt_classify_sharded_device",
    "ddt_group_create", "ddt_group_destroy", "ddt_group_last_error", "ddt_group_engine", "ddt_group_load_model",
    "ddt_group_load_model_sparse", "ddt_group_score", "ddt_group_load_model_multiclass", "ddt_group_classify", "ddt_debug_prepass_image", "ddt_debug_sparse_image", "ddt_debug_rank32_tables",
    "ddt_shard_range", "ddt_debug_model_image", "ddt_comm_chunk_schedule", "ddt_comm_score", "ddt_group_load_model_replicated", "ddt_group_score_rows",
    "ddt_host_register", "ddt_host_unregister",
    "ddt_comm_create_hybrid", "ddt_comm_layout", "ddt_hybrid_rows", "ddt_score_hybrid_device", "ddt_classify_hybrid_device", "ddt_comm_abort",
    "ddt_group_create_hybrid",
]


class Params(C.Structure):
    """ddt_params (include/ddt.h) == the reference's CSR 204/205 fields (EngineCSR.sv:218-233)."""

    _fields_ = [
        ("num_trees", C.c_uint32), ("num_levels", C.c_uint32), ("num_features", C.c_uint32),
        ("missing_bits", C.c_uint32), ("weights_lines_per_tree", C.c_uint32),
        ("findex_lines_per_tree", C.c_uint32), ("cmp_mode", C.c_uint32), ("clusters_per_tuple", C.c_uint32),
        ("sum_mode", C.c_uint32), ("reserved", C.c_uint32 * 3),
    ]


class CommLayout(C.Structure):
    """ddt_comm_layout_t (include/ddt.h)"""

    _fields_ = [("rank", C.c_int), ("n_ranks", C.c_int), ("tree_ranks", C.c_int), ("tree_rank", C.c_int), ("row_groups", C.c_int), ("row_group", C.c_int)]


class Info(C.Structure):
    _fields_ = [
        ("abi_version", C.c_uint32), ("device_id", C.c_int32), ("tree_begin", C.c_uint32), ("tree_end", C.c_uint32),
        ("num_levels", C.c_uint32), ("num_features", C.c_uint32), ("tuple_words", C.c_uint32),
        ("variant", C.c_uint32), ("tile_tuples", C.c_uint32), ("block_threads", C.c_uint32),
        ("lds_bytes", C.c_uint32), ("num_classes", C.c_uint32), ("local_trees", C.c_uint32),
        ("model_bytes_unpadded", C.c_uint64), ("image_bytes", C.c_uint64),
        ("variant_name", C.c_char * 64), ("device_name", C.c_char * 64),
        ("num_cus", C.c_uint32), ("clock_khz", C.c_uint32), ("lds_bytes_per_cu", C.c_uint32), ("prepass_groups", C.c_uint32),
        ("fallback_kernel", C.c_uint32), ("build_checks", C.c_uint32),
    ]


class Stats(C.Structure):
    _fields_ = [
        ("tuples_in", C.c_uint64), ("tuples_out", C.c_uint64), ("tuple_lines_in", C.c_uint64),
        ("result_lines_out", C.c_uint64), ("model_lines_in", C.c_uint64), ("score_calls", C.c_uint64),
        ("kernel_launches", C.c_uint64), ("prog_ms", C.c_double), ("exec_ms", C.c_double),
        ("last_prepass_ms", C.c_double), ("last_score_ms", C.c_double),
        ("timed_launches", C.c_uint64), ("sum_prepass_ms", C.c_double), ("sum_score_ms", C.c_double),
    ]


def build(force: bool = False) -> str:
    """Compile csrc/ for gfx950 with hipcc (cross-compiles without a GPU)."""
    args = ["make", "-C", CSRC, "-s"]
    if force:
        subprocess.check_call(args + ["clean"])
    subprocess.check_call(args)
    return LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} is missing: build it with `make -C {CSRC}` (or __graft_entry__.build()). "
            "The scoring path has no fallback.")
    # torch bundles its own HIP runtime (same soname); import it first so that libddt.so binds to the
    # runtime that owns torch's device allocations and streams.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is optional for the pure C-ABI use
        pass
    _lib = bind(C.CDLL(LIB_PATH))
    return _lib


def bind(L):
    """Set the prototypes of include/ddt.h on a loaded library (libddt.so; tests also bind their CPU-model build of the host side)."""
    vp, sz, u32, u64, i32, i64 = C.c_void_p, C.c_size_t, C.c_uint32, C.c_uint64, C.c_int, C.c_int64
    PP = C.POINTER(Params)
    L.ddt_create.restype, L.ddt_create.argtypes = i32, [C.POINTER(vp), i32]
    L.ddt_destroy.restype, L.ddt_destroy.argtypes = None, [vp]
    L.ddt_load_model.restype, L.ddt_load_model.argtypes = i32, [vp, PP, vp, sz, vp, sz]
    L.ddt_load_model_shard.restype, L.ddt_load_model_shard.argtypes = i32, [vp, PP, vp, sz, vp, sz, u32, u32]
    L.ddt_score.restype, L.ddt_score.argtypes = i32, [vp, vp, sz, vp]
    L.ddt_score_device.restype, L.ddt_score_device.argtypes = i32, [vp, vp, sz, vp, vp]
    L.ddt_host_register.restype, L.ddt_host_register.argtypes = i32, [vp, vp, sz]
    L.ddt_host_unregister.restype, L.ddt_host_unregister.argtypes = i32, [vp, vp]
    L.ddt_chain_sum_device.restype, L.ddt_chain_sum_device.argtypes = i32, [vp, vp, u32, sz, vp, vp]
    L.ddt_load_model_multiclass.restype = i32
    L.ddt_load_model_multiclass.argtypes = [vp, PP, vp, sz, vp, sz, u32, i32, u32, u32]
    L.ddt_classify_device.restype, L.ddt_classify_device.argtypes = i32, [vp, vp, sz, vp, vp, vp]
    L.ddt_classify.restype, L.ddt_classify.argtypes = i32, [vp, vp, sz, vp, vp]
    L.ddt_argmax_device.restype, L.ddt_argmax_device.argtypes = i32, [vp, vp, u32, sz, vp, vp]
    L.ddt_csr_encode.restype, L.ddt_csr_encode.argtypes = i32, [PP, u64, u32, C.POINTER(u64 * 12)]
    L.ddt_csr_decode.restype = i32
    L.ddt_csr_decode.argtypes = [C.POINTER(u64 * 12), PP, C.POINTER(u64), C.POINTER(u32)]
    L.ddt_get_info.restype, L.ddt_get_info.argtypes = i32, [vp, C.POINTER(Info)]
    L.ddt_get_stats.restype, L.ddt_get_stats.argtypes = i32, [vp, C.POINTER(Stats)]
    L.ddt_strerror.restype, L.ddt_strerror.argtypes = C.c_char_p, [i32]
    L.ddt_last_error.restype, L.ddt_last_error.argtypes = C.c_char_p, [vp]
    L.ddt_set_option.restype, L.ddt_set_option.argtypes = i32, [vp, C.c_char_p, i64]
    L.ddt_num_variants.restype, L.ddt_num_variants.argtypes = i32, []
    L.ddt_variant_name.restype, L.ddt_variant_name.argtypes = i32, [i32, C.c_char_p, sz]
    L.ddt_load_model_sparse.restype, L.ddt_load_model_sparse.argtypes = i32, [vp, PP, vp, sz, vp, u32, u32]
    L.ddt_load_model_sparse_multiclass.restype = i32
    L.ddt_load_model_sparse_multiclass.argtypes = [vp, PP, vp, sz, vp, u32, i32, u32, u32]
    L.ddt_synth_sparse_model.restype = C.c_int64
    L.ddt_synth_sparse_model.argtypes = [u32, u32, u32, u32, u32, i32, vp, sz, vp]
    L.ddt_csr_encode_ex.restype, L.ddt_csr_encode_ex.argtypes = i32, [PP, u64, u32, u32, u32, C.POINTER(u64 * 12)]
    L.ddt_csr_decode_ex.restype = i32
    L.ddt_csr_decode_ex.argtypes = [C.POINTER(u64 * 12), PP, C.POINTER(u64), C.POINTER(u32), C.POINTER(u32),
                                    C.POINTER(u32), C.POINTER(C.c_uint8 * 20)]
    L.ddt_debug_model_image.restype = i32
    L.ddt_debug_model_image.argtypes = [PP, vp, sz, vp, sz, i32, vp, vp, sz, vp, sz, C.POINTER(u64 * 12)]
    L.ddt_group_load_model_replicated.restype, L.ddt_group_load_model_replicated.argtypes = i32, [vp, PP, vp, sz, vp, sz]
    L.ddt_group_score_rows.restype, L.ddt_group_score_rows.argtypes = i32, [vp, vp, sz, vp]
    L.ddt_comm_score.restype, L.ddt_comm_score.argtypes = i32, [vp, vp, sz, vp, i32]
    L.ddt_comm_chunk_schedule.restype, L.ddt_comm_chunk_schedule.argtypes = C.c_int64, [sz, sz, i32, sz, vp, sz]
    L.ddt_shard_range.restype, L.ddt_shard_range.argtypes = i32, [u32, u32, u32, C.POINTER(u32), C.POINTER(u32)]
    # multi-GPU jobs (RCCL behind the C-ABI)
    L.ddt_comm_get_unique_id.restype, L.ddt_comm_get_unique_id.argtypes = i32, [vp]
    L.ddt_comm_create.restype, L.ddt_comm_create.argtypes = i32, [C.POINTER(vp), vp, i32, i32, vp]
    L.ddt_comm_destroy.restype, L.ddt_comm_destroy.argtypes = None, [vp]
    L.ddt_comm_last_error.restype, L.ddt_comm_last_error.argtypes = C.c_char_p, [vp]
    L.ddt_comm_set_option.restype, L.ddt_comm_set_option.argtypes = i32, [vp, C.c_char_p, i64]
    L.ddt_score_sharded_device.restype, L.ddt_score_sharded_device.argtypes = i32, [vp, vp, sz, vp, i32, vp]
    L.ddt_score_rowsharded_device.restype, L.ddt_score_rowsharded_device.argtypes = i32, [vp, vp, sz, vp, vp]
    L.ddt_classify_sharded_device.restype, L.ddt_classify_sharded_device.argtypes = i32, [vp, vp, sz, vp, vp, i32, vp]
    L.ddt_comm_create_hybrid.restype, L.ddt_comm_create_hybrid.argtypes = i32, [C.POINTER(vp), vp, i32, i32, i32, vp]
    L.ddt_comm_layout.restype, L.ddt_comm_layout.argtypes = i32, [vp, C.POINTER(CommLayout)]
    L.ddt_hybrid_rows.restype, L.ddt_hybrid_rows.argtypes = i32, [sz, i32, i32, C.POINTER(sz), C.POINTER(sz)]
    L.ddt_score_hybrid_device.restype, L.ddt_score_hybrid_device.argtypes = i32, [vp, vp, sz, vp, i32, i32, vp]
    L.ddt_classify_hybrid_device.restype, L.ddt_classify_hybrid_device.argtypes = i32, [vp, vp, sz, vp, vp, i32, i32, vp]
    L.ddt_comm_abort.restype, L.ddt_comm_abort.argtypes = i32, [vp]
    L.ddt_group_create_hybrid.restype, L.ddt_group_create_hybrid.argtypes = i32, [C.POINTER(vp), i32, C.POINTER(i32), i32]
    L.ddt_group_create.restype, L.ddt_group_create.argtypes = i32, [C.POINTER(vp), i32, C.POINTER(i32)]
    L.ddt_group_destroy.restype, L.ddt_group_destroy.argtypes = None, [vp]
    L.ddt_group_last_error.restype, L.ddt_group_last_error.argtypes = C.c_char_p, [vp]
    L.ddt_group_engine.restype, L.ddt_group_engine.argtypes = vp, [vp, i32]
    L.ddt_group_load_model.restype, L.ddt_group_load_model.argtypes = i32, [vp, PP, vp, sz, vp, sz]
    L.ddt_group_load_model_sparse.restype, L.ddt_group_load_model_sparse.argtypes = i32, [vp, PP, vp, sz, vp]
    L.ddt_group_score.restype, L.ddt_group_score.argtypes = i32, [vp, vp, sz, vp, i32]
    L.ddt_group_load_model_multiclass.restype, L.ddt_group_load_model_multiclass.argtypes = i32, [vp, PP, vp, sz, vp, sz, u32, i32]
    L.ddt_group_classify.restype, L.ddt_group_classify.argtypes = i32, [vp, vp, sz, vp, vp, i32]
    L.ddt_debug_prepass_image.restype, L.ddt_debug_prepass_image.argtypes = C.c_int64, [vp, vp, u32, u32, vp, sz, vp]
    L.ddt_debug_sparse_image.restype, L.ddt_debug_sparse_image.argtypes = i32, [PP, vp, sz, vp, i32, i32, vp, sz, vp, sz, vp]
    L.ddt_synth_model.restype, L.ddt_synth_model.argtypes = i32, [u32, u32, u32, i32, vp, vp]
    L.ddt_synth_tuples_host.restype, L.ddt_synth_tuples_host.argtypes = i32, [vp, u64, sz, u32, i32, u32]
    L.ddt_synth_tuples_device.restype, L.ddt_synth_tuples_device.argtypes = i32, [vp, vp, u64, sz, u32, i32, u32, vp]
    return L
