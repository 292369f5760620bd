"""Host half of the model load, checked without a GPU: the packed device images of perfect-tree models.

ddt_debug_model_image (include/ddt.h, host-only) validates and parses the two model streams exactly as ddt_load_model does
and returns the image the chosen kernel variant consumes (csrc/ddt_internal.h: 8-byte node records + leaves, the fused
last-level layout, the rank-quantised 4-byte records incl. the `_gl` chunk layout, the image for tiles with missing values,
the per-feature threshold tables, EMPTY padding trees).  The walks below address those images the way the kernels do
(ddt_kernels.hip walk_trees / walk_trees_q16: 1-based heap, record -> feature row -> compare -> child, leaf read) in numpy
and must select, for every (tuple, tree), the leaf the oracle's walk of the WIRE format selects (orc_leaves, itself pinned
to the reference's RTL: tests/test_oracle_adder.py, tests/test_oracle_program.py).  The GPU parity tests then only have to
show that the kernels read these images as described."""
import ctypes as C

import numpy as np
import pytest

import ddt
from oracle import oracle as O

GENERIC, TILE, STREAM, Q16 = 0, 1, 2, 3


def _variant(name):
    names = ddt.variant_names()
    assert name in names, name
    return names.index(name)


def _image(p, w, f, variant):
    L = ddt.lib()
    info = (C.c_uint64 * 12)()
    w = np.ascontiguousarray(w, np.uint32).reshape(-1)
    f = np.ascontiguousarray(f, np.uint16).reshape(-1)
    rc = L.ddt_debug_model_image(C.byref(p), w.ctypes.data, w.size // 4, f.ctypes.data, f.size // 8, variant, None, None, 0, None, 0, C.byref(info))
    assert rc == 0, rc
    img = np.zeros(info[0], np.uint32)
    slow = np.zeros(info[0] if info[2] == Q16 else 0, np.uint32)
    tab = np.zeros(info[11], np.uint32)
    rc = L.ddt_debug_model_image(C.byref(p), w.ctypes.data, w.size // 4, f.ctypes.data, f.size // 8, variant, img.ctypes.data,
                                 slow.ctypes.data if slow.size else None, img.size, tab.ctypes.data if tab.size else None, tab.size, C.byref(info))
    assert rc == 0, rc
    keys = ("words", "Tpad", "kind", "opt", "chunk_trees", "tile", "feat_off", "row", "Kpad", "W", "variant", "table_words")
    nfo = dict(zip(keys, (int(x) for x in info)))
    nfo["T"], nfo["clusters"] = int(p.num_trees), int(p.clusters_per_tuple)  # what the cluster-major ("_cm") image order depends on
    return img, slow, tab, nfo


def _ieee_key(b):
    b = b.astype(np.uint32)
    k = np.where(b & 0x80000000, b ^ 0x7FFFFFFF, b)
    k = np.where(b == 0x80000000, 0, k)
    return np.where((b & 0x7FFFFFFF) > 0x7F800000, 0x7FFFFFFF, k).astype(np.uint32)


def _walk_records(img, nfo, D, x, missing, cmp_mode):
    """Tile / stream / generic images (ddt_internal.h layouts 0 and 1) -> leaf bits [n, Tpad]."""
    n = x.shape[0]
    rows = np.arange(n)
    keys = (x if cmp_mode == 0 else _ieee_key(x)).view(np.int32)
    miss = x == missing
    tw = (12 << D) // 4
    fused = nfo["kind"] == TILE and (nfo["opt"] & 1)
    out = np.zeros((n, nfo["Tpad"]), np.uint32)
    for i in range(nfo["Tpad"]):
        t = img[i * tw:(i + 1) * tw]
        m = np.ones(n, np.int64)
        leaf = None
        for lvl in range(D):
            last = fused and lvl == D - 1
            if last:  # {thr, w2, leafL, leafR} at 4*2^D + 16*(m - 2^(D-1))
                base = (4 << D) // 4 + 4 * (m - (1 << (D - 1)))
                key, word = t[base], t[base + 1]
            else:
                key, word = t[2 * m], t[2 * m + 1]
            addr = word & 0x7FFFFFFF
            if nfo["kind"] == GENERIC:
                j = addr
            else:
                j = (addr - nfo["feat_off"]) // nfo["row"]
                # stream kernels: the rows of tuple line q start 64 * q bytes late (Variant::feat_word_stream)
                skew = 64 * (j // 4) if nfo["kind"] == STREAM else 0
                assert (addr - nfo["feat_off"] == j * nfo["row"] + skew).all()
            right = np.where(miss[rows, j], word >> 31, ~(keys[rows, j] < key.view(np.int32))).astype(np.int64)
            if last:
                leaf = np.where(right == 1, t[base + 3], t[base + 2])
            m = 2 * m + right
        out[:, i] = leaf if fused else t[(8 << D) // 4 + m - (1 << D)]
    return out


def _ranks(tab, nfo, x, missing, cmp_mode):
    """rank of every feature value among its feature's sorted threshold keys; missing -> 0xFFFF (kQMissing)."""
    keys = (x if cmp_mode == 0 else _ieee_key(x)).view(np.int32)
    T = tab.reshape(nfo["W"], nfo["Kpad"]).view(np.int32)
    assert (np.diff(T.astype(np.int64), axis=1) >= 0).all()          # sorted, INT_MAX pads behind
    r = np.stack([np.searchsorted(T[j], keys[:, j], side="right") for j in range(nfo["W"])], axis=1)
    return np.where(x == missing, 0xFFFF, r).astype(np.int64)


def _walk_q16(img, nfo, D, rank, slow):
    """Rank-quantised images: 4-byte records {R | row offset << 16 (| miss_right << 16 in the slow image)} -> leaf bits."""
    n = rank.shape[0]
    rows = np.arange(n)
    half, CT, row = 1 << D, nfo["chunk_trees"], nfo["row"]
    tw = 2 * half
    gl = nfo["opt"] & 1
    out = np.zeros((n, nfo["Tpad"]), np.uint32)
    for tree in range(nfo["Tpad"]):
        i = tree
        g, C_, G = tree // 8, nfo["clusters"], (nfo["T"] + 7) // 8
        if nfo["opt"] & 4 and g < G:  # "_cm": the real PU groups in cluster-major order (cluster = group % C), each cluster's in their order
            i = (sum((G + C_ - 1 - k) // C_ for k in range(g % C_)) + g // C_) * 8 + tree % 8
        rec_off = (i // CT) * CT * tw + (i % CT) * half if gl else i * tw
        leaf_off = (i // CT) * CT * tw + CT * half + (i % CT) * half if gl else i * tw + half
        m = np.ones(n, np.int64)
        for _ in range(D):
            rec = img[rec_off + m]
            off = (rec >> 16) & (0xFFFE if slow else 0xFFFF)
            assert (off % row == 0).all()
            f = rank[rows, off // row]
            right = f >= (rec & 0xFFFF)
            if slow:
                right = np.where(f == 0xFFFF, ((rec >> 16) & 1) != 0, right)
            m = 2 * m + right.astype(np.int64)
        out[:, tree] = img[leaf_off + m - half]
    return out


def _check(T, D, F, variant_name, dist, cmp_mode=0, n=96, expect_auto=None):
    m = O.gen_model(T, D, F, dist, cmp_mode=cmp_mode)
    p = ddt.make_params(T, D, F, cmp_mode=cmp_mode)
    x = O.gen_tuples(0, n, F, dist)
    if dist:  # values exactly on a threshold, on both sides of zero
        nint = (1 << D) - 1
        thr = m.wlines.reshape(T, -1)[:, :nint]
        fid = m.flines.reshape(T, -1)[:, :nint] & 0x7FF
        for k in range(min(n, 16)):
            x[k, fid[k % T, k % nint]] = thr[k % T, k % nint]
    missing = p.missing_bits
    want = np.stack([O.leaves(m, row) for row in x])
    img, slow, tab, nfo = _image(p, m.wlines, m.flines, -1 if variant_name is None else _variant(variant_name))
    if expect_auto is not None:
        assert ddt.variant_names()[nfo["variant"]] == expect_auto
    assert nfo["Tpad"] >= T and nfo["W"] == (F + 3) // 4 * 4
    if nfo["kind"] == Q16:
        rank = _ranks(tab, nfo, x, missing, cmp_mode)
        got = _walk_q16(slow, nfo, D, rank, True)                      # the image tiles with a missing value use
        clean = ~(x == missing).any(axis=1)
        if clean.any():                                                # tiles without one: no flags, no masking
            fast = _walk_q16(img, nfo, D, rank[clean], False)
            assert np.array_equal(fast[:, :T], want[clean]), "fast image"
    else:
        got = _walk_records(img, nfo, D, x, missing, cmp_mode)
    assert np.array_equal(got[:, :T], want), (variant_name, T, D, F)
    assert not got[:, T:].any(), "EMPTY padding trees select +0"
    return nfo


def test_headline_model_takes_the_gl_rank_quantised_image():
    nfo = _check(1000, 8, 32, None, 0, n=48, expect_auto="q16_d8_c8_u4_gl_s2_cm_x")   # cluster-major image, no accumulator ring, pinned read order
    assert nfo["kind"] == Q16 and nfo["opt"] & 1 and nfo["chunk_trees"] == 8 and nfo["Tpad"] == 1000 and nfo["tile"] == 1024


def test_eight_way_shard_of_the_headline_model():
    nfo = _check(125, 8, 32, None, 0, n=48, expect_auto="q16_d8_c8_u4_gl_s2_cm_x")
    assert nfo["Tpad"] == 128                                          # whole chunks of 8: three EMPTY trees


@pytest.mark.parametrize("name,T,D,F", [("q16_d8_c8_u4_gl_s2_cm_x", 37, 8, 32), ("q16_d8_c8_u4_gl_s2_cm_x", 300, 8, 32), ("q16_d8_c8_u4_gl_s2_cm_x", 1000, 8, 20), ("q16_d8_c8_u4_gl_s2", 37, 8, 32), ("q16_d8_c8_u4_gl", 37, 8, 32), ("q16_d7_c8_u4", 37, 7, 32), ("q16_d6_c16_u4", 100, 6, 28),
                                        ("q16_d4_c64_u8", 9, 4, 16), ("q16_d3_c128_u8", 130, 3, 7)])
def test_rank_quantised_images_with_missing_values(name, T, D, F):
    _check(T, D, F, name, 1)


@pytest.mark.parametrize("name,T,D,F", [("d8_t1024_r1_c4_u4_dma_f", 37, 8, 32), ("d8_t512_r1_c4_u4_dma_f", 12, 8, 64), ("d8_t512_r1_c8_u8_dma_f", 20, 8, 40),
                                        ("d6_t1024_r1_c16_u4_dma", 100, 6, 28), ("d4_t256_r1_c64_u8_dma", 70, 4, 100),
                                        ("stream_d4_u4_l4", 8, 4, 16), ("stream_d6_u4_l8", 3, 6, 30)])
def test_tile_and_stream_images(name, T, D, F):
    _check(T, D, F, name, 1)


@pytest.mark.parametrize("T,D,F", [(5, 11, 64), (9, 2, 5), (3, 8, 300)])
def test_generic_images(T, D, F):
    nfo = _check(T, D, F, ddt.variant_names()[0], 1)
    assert nfo["kind"] == GENERIC


@pytest.mark.parametrize("name", ["q16_d8_c8_u4_gl", "d8_t1024_r1_c4_u4_dma_f"])
def test_ieee_compare_mode(name):
    _check(21, 8, 32, name, 1, cmp_mode=1)


def test_config_2_and_1_choices():
    assert ddt.variant_names()[_check(100, 6, 28, None, 0, n=32)["variant"]] == "q16_d6_c16_u4_s2"   # config 2: 600 tree-levels, above the q16 break-even since round 3
    assert ddt.variant_names()[_check(40, 6, 28, None, 0, n=32)["variant"]].startswith("d6_t1024")      # small ensembles stay on the fp32 tile kernel
    assert ddt.variant_names()[_check(8, 4, 16, None, 0, n=32)["variant"]].startswith("stream_d4")


def test_hook_refuses_what_the_loader_refuses():
    L = ddt.lib()
    m = O.gen_model(8, 4, 16, 0)
    p = ddt.make_params(8, 4, 16)
    info = (C.c_uint64 * 12)()
    args = (m.wlines.ctypes.data, m.wlines.size // 4, m.flines.ctypes.data, m.flines.size // 8)
    assert L.ddt_debug_model_image(C.byref(p), *args, _variant("q16_d8_c8_u4_gl"), None, None, 0, None, 0, C.byref(info)) == -5  # wrong depth
    assert L.ddt_debug_model_image(C.byref(p), args[0], 1, args[2], args[3], -1, None, None, 0, None, 0, C.byref(info)) == -1     # stream too short
    bad = m.flines.copy()
    bad[0] = 16                                                        # feature index >= F
    assert L.ddt_debug_model_image(C.byref(p), args[0], args[1], bad.ctypes.data, args[3], -1, None, None, 0, None, 0, C.byref(info)) == -1


def test_a_rank_table_longer_than_32767_keys_is_padded_to_a_multiple_of_32():
    """One feature carrying 140 x 255 thresholds (~35.7 k distinct): since round 6 one table -- what a block of rank_kernel holds in 160 KiB of
    LDS (csrc/ddt_engine_priv.h kQ16MaxTable = 38848) -- padded to a multiple of 32 entries instead of the next power of two; the image's
    records then carry ranks above 32767 and walk to the oracle's leaves like any other."""
    nfo = _check(140, 8, 1, None, 1, n=40, expect_auto="q16_d8_c8_u4_gl_s2_cm_x")
    assert nfo["kind"] == Q16 and 32768 < nfo["Kpad"] <= 38912 and nfo["Kpad"] % 32 == 0
    nfo = _check(100, 8, 1, None, 0, n=16)                                          # 25.5 k keys: a power of two as before
    assert nfo["Kpad"] == 32768
