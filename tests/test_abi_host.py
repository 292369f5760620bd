"""CPU-side checks of the product library: it loads, exports exactly the symbols include/ddt.h declares,
links nothing from oracle/, refuses to run without a GPU (no CPU fallback), and its host-side pieces
(synthetic generators, parameter validation surface) agree with the oracle bit for bit."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

from oracle import oracle as O
import ddt
from ddt import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    src = open(os.path.join(ROOT, "include", "ddt.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ddt_[a-z_0-9]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    assert os.path.exists(_lib.LIB_PATH), "libddt.so not built (run __graft_entry__.build())"
    L = C.CDLL(_lib.LIB_PATH)
    declared = _header_symbols()
    assert declared == sorted(_lib.SYMBOLS)
    for s in declared:
        assert hasattr(L, s), s


def test_product_does_not_link_the_oracle():
    out = subprocess.check_output(["nm", "-D", "--defined-only", _lib.LIB_PATH]).decode()
    assert "orc_" not in out
    und = subprocess.check_output(["nm", "-D", "--undefined-only", _lib.LIB_PATH]).decode()
    assert "orc_" not in und
    needed = subprocess.check_output(["readelf", "-d", _lib.LIB_PATH]).decode()
    assert "liboracle" not in needed
    for root, _, files in os.walk(os.path.join(ROOT, "distributed-decisiontrees_amd")):
        for f in files:
            if f.endswith((".py", ".cpp", ".hip", ".h")):
                txt = open(os.path.join(root, f)).read()
                assert "liboracle" not in txt and "import oracle" not in txt and "from oracle" not in txt, f


def test_kernels_have_no_static_lds_and_no_scratch():
    # absolute LDS addressing in ddt_kernels.hip relies on the dynamic segment starting at address 0
    import tempfile

    llvm = "/opt/rocm/lib/llvm/bin"
    with tempfile.TemporaryDirectory() as td:
        fat, co = os.path.join(td, "fat.bin"), os.path.join(td, "dev.co")
        subprocess.check_call(["objcopy", "-O", "binary", "--only-section=.hip_fatbin", _lib.LIB_PATH, fat])
        subprocess.check_call([f"{llvm}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={fat}",
                               "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"])
        out = subprocess.check_output([f"{llvm}/llvm-readelf", "--notes", co]).decode(errors="ignore")
    names = re.findall(r"\.name:\s+(\S+)", out)
    fixed = re.findall(r"\.group_segment_fixed_size:\s+(\d+)", out)
    scratch = re.findall(r"\.private_segment_fixed_size:\s+(\d+)", out)
    assert len(fixed) >= 10 and len(fixed) == len(scratch) == len(names)
    assert sum("score_tile_kernel" in n for n in names) >= 8  # gfx950 code objects for the tile kernels
    assert all(int(x) == 0 for x in fixed), "static LDS found: dynamic LDS no longer starts at 0"
    assert all(int(x) == 0 for x in scratch), "a kernel spills to scratch"


def test_no_gpu_means_error_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    h = C.c_void_p()
    rc = ddt.lib().ddt_create(C.byref(h), 0)
    assert rc == -6 and not h.value  # DDT_ENODEVICE
    assert b"no CPU fallback" in ddt.lib().ddt_strerror(rc)
    with pytest.raises(ddt.DDTError):
        ddt.Engine(0)


@pytest.mark.parametrize("T,D,F,dist", [(8, 4, 16, 0), (100, 6, 28, 1), (13, 8, 32, 0), (2, 12, 100, 1)])
def test_synthetic_generators_match_oracle_bit_for_bit(T, D, F, dist):
    w, f = ddt.synth_model(T, D, F, dist)
    m = O.gen_model(T, D, F, dist=dist)
    assert np.array_equal(w, m.wlines) and np.array_equal(f, m.flines)
    x = ddt.synth_tuples_host(123456789, 300, F, dist)
    assert np.array_equal(x, O.gen_tuples(123456789, 300, F, dist=dist))


def test_variant_table_and_helpers():
    names = ddt.variant_names()
    assert names[0] == "generic" and len(names) == ddt.lib().ddt_num_variants() and len(set(names)) == len(names)
    assert ddt.weights_lines_per_tree(8) == 128 and ddt.findex_lines_per_tree(8) == 32
    assert ddt.weights_lines_per_tree(4) == 8 and ddt.findex_lines_per_tree(4) == 2
    assert ddt.tuple_words(28) == 28 and ddt.tuple_words(30) == 32
    assert [ddt.default_clusters(t) for t in (1, 128, 129, 256, 257, 512, 513, 1000)] == [1, 1, 2, 2, 4, 4, 8, 8]
    p = ddt.make_params(1000, 8, 32)
    assert C.sizeof(p) == 48 and p.clusters_per_tuple == 8


def test_shard_bounds_match_oracle_split():
    from tests import refimpl as R

    for T, G in [(1000, 8), (1000, 3), (10, 4), (7, 7)]:
        assert ddt.shard_bounds(T, G) == R.shard_bounds(T, G)


def test_golden_fixtures_still_match_the_oracle():
    gdir = os.path.join(ROOT, "tests", "golden")
    files = sorted(f for f in os.listdir(gdir) if f.endswith(".npz") and not f.endswith("_rtl_vectors.npz"))
    assert len(files) >= 4
    for fn in files:
        g = np.load(os.path.join(gdir, fn))
        p = O.Params(*[int(v) for v in g["params"]])
        m = O.Model(p, g["wlines"], g["flines"])
        assert np.array_equal(O.score(m, g["tuples"]).view(np.uint32), g["score_ref_bits"]), fn
        assert np.array_equal(O.score(m, g["tuples"], sum_mode=O.SUM_REF_NATIVE).view(np.uint32), g["score_ref_bits"])
        assert np.array_equal(O.score(m, g["tuples"], sum_mode=O.SUM_F64_SEQ).view(np.uint32), g["score_f64_bits"])
        assert np.array_equal(O.score(m, g["tuples"], n_devices=2).view(np.uint32), g["score_ref_2dev_bits"])
