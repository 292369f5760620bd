"""ddt -- host-side Python mirror of libddt.so (MI355X-native decision-tree-ensemble scoring).

The product is the HIP library behind the C-ABI in include/ddt.h; this package is plumbing (ctypes +
torch device memory / streams / torch.distributed)."""
from ._lib import Info, Params, Stats, build, lib  # noqa: F401
from .engine import (COMBINE_ALLREDUCE, COMBINE_CHAIN, Comm, DDTError, Engine, Group, comm_unique_id, hybrid_rows, default_clusters, findex_lines_per_tree, make_params,  # noqa: F401
                     make_sparse_params, shard_bounds, synth_model, synth_sparse_model, synth_tuples_host, tuple_words, variant_names, weights_lines_per_tree)
from . import importer  # noqa: F401

CLI_PATH = __import__("os").path.join(__import__("os").path.dirname(__import__("os").path.dirname(__file__)), "bin", "ddt_cli")
