"""Host logic of the sparse (explicit-children) path, no GPU needed: the device images `ddt_sparse_host.cpp sparse_pack_host`
builds (top-K perfect heap images per PU group with dummy nodes under early leaves, level K-1 records, the deep record
array in level or depth-first order) come back through the test hook `ddt_debug_sparse_image` and
are walked in numpy the way `score_sparse_kernel` walks them; every (tuple, tree) must end on the leaf the oracle's walk of
the wire format ends on, EMPTY slots on +0."""
import ctypes as C
import re

import numpy as np
import pytest

from oracle import oracle as O
import ddt
from ddt import _lib

LEFT_LEAF, RIGHT_LEAF, MISS_RIGHT, ADDR = 0x80000000, 0x40000000, 0x20000000, 0x1FFFFFFF


def _images(s, variant, order):
    L = _lib.lib()
    nl = np.ascontiguousarray(s.node_lines).view(np.uint32).reshape(-1, 4)
    first = np.ascontiguousarray(s.first, dtype=np.uint64)
    q = s.params
    p = ddt.make_sparse_params(q.num_trees, q.num_levels, q.num_features, q.missing_bits, q.cmp_mode, q.clusters_per_tuple, 0)
    info = np.zeros(6, np.uint64)
    rc = L.ddt_debug_sparse_image(C.byref(p), nl.ctypes.data, nl.shape[0], first.ctypes.data, variant, order, None, 0, None, 0, info.ctypes.data)
    assert rc == 0, rc
    top, deep = np.zeros(int(info[0]), np.uint32), np.zeros(int(info[1]), np.uint32)
    rc = L.ddt_debug_sparse_image(C.byref(p), nl.ctypes.data, nl.shape[0], first.ctypes.data, variant, order, top.ctypes.data, top.size,
                                  deep.ctypes.data, deep.size, info.ctypes.data)
    assert rc == 0, rc
    return top, deep.reshape(-1, 4), [int(v) for v in info]


def _rank_tables(s):
    """per feature: the sorted distinct threshold keys of the forest (cmp_mode 0: the raw bits in int32 order)"""
    nl = np.ascontiguousarray(s.node_lines).view(np.uint32).reshape(-1, 4)
    return [np.unique(nl[(nl[:, 1] & 0x7FF) == j, 0].view(np.int32)) for j in range(int(s.params.num_features))]


def _walk(top, deep, info, slot, x, miss_bits, tables=None, mid=0, pairs=False):
    """score_sparse_kernel's walk of one tree slot for one tuple (cmp_mode 0: signed compare of the raw bits); `tables`
    (rank-quantised kernels): the node word is the threshold's rank, the feature value is replaced by ITS rank = number of keys <= x"""
    _, _, _, K, feat_off, row = info
    per_tree = top.size // (info[2] * 8)  # 12 * 2^K bytes, or 8 * 2^K (dense level K: all K levels as 8-byte records)
    t = top[slot * per_tree: (slot + 1) * per_tree]
    dense = per_tree == (8 << K) // 4

    def right(key, w):
        j = ((w & ADDR) - feat_off) // row
        f = int(x[j])
        if f == miss_bits:
            return (w & MISS_RIGHT) != 0
        if tables is not None:
            return int(np.searchsorted(tables[j], np.uint32(f).view(np.int32), side="right")) >= key if j < len(tables) else 0 >= key
        return np.int32(np.uint32(f).view(np.int32)) >= np.uint32(key).view(np.int32)

    m = 1
    for _ in range(K if dense else K - 1):
        m = 2 * m + int(right(int(t[2 * m]), int(t[2 * m + 1])))
    if dense and pairs:  # dense pair records ("sparse_dp_*"): one 16-byte record {key, left key, right key, feature numbers as bytes + missing
        # directions in bits 24..26} per level-K node at byte cbase + 16 h; the dense block of level K + 2 at byte cbase - 32 * 2^K + 16 h

        def right_j(key, j, miss_right):
            f = int(x[j])
            if f == miss_bits:
                return bool(miss_right)
            if tables is not None:  # rank-quantised: the key is the threshold's rank, the feature its own rank
                return int(np.searchsorted(tables[j], np.uint32(f).view(np.int32), side="right")) >= key if j < len(tables) else 0 >= key
            return bool(np.uint32(f).view(np.int32) >= np.uint32(key).view(np.int32))

        off = (16 * m + int(t[0])) & 0xFFFFFFFF
        assert off % 16 == 0 and off // 16 < deep.shape[0]
        ka, kl, kr, w = (int(v) for v in deep[off // 16])
        r0 = right_j(ka, w & 0xFF, (w >> 24) & 1)
        r1 = right_j(kr if r0 else kl, (w >> (16 if r0 else 8)) & 0xFF, (w >> (26 if r0 else 25)) & 1)
        m = 4 * m + 2 * int(r0) + int(r1)
        off = (16 * m + int(t[0]) - (32 << K)) & 0xFFFFFFFF
        assert off % 16 == 0 and off // 16 < deep.shape[0]
        rec = deep[off // 16]
    elif dense and mid:  # dense mid levels ("sparse_dm<M>_*"): M levels of 8-byte records that continue the heap at byte cbase + 8 h, then the
        # dense block of 16-byte records of level K + M at byte cbase - 8 * 2^(K+M) + 16 h
        words = deep.reshape(-1)
        for _ in range(mid):
            off = (8 * m + int(t[0])) & 0xFFFFFFFF
            assert off % 8 == 0 and off // 4 + 1 < words.size
            m = 2 * m + int(right(int(words[off // 4]), int(words[off // 4 + 1])))
        off = (16 * m + int(t[0]) - (8 << (K + mid))) & 0xFFFFFFFF
        assert off % 16 == 0 and off // 16 < deep.shape[0]
        rec = deep[off // 16]
    elif dense:  # level K: deep record at byte 2 * (8 m) + cbase (mod 2^32), cbase in word 0 of the tree's image
        off = (16 * m + int(t[0])) & 0xFFFFFFFF
        assert off % 16 == 0 and off // 16 < deep.shape[0]
        rec = deep[off // 16]
    else:
        rec = t[(4 << K) // 4 + 4 * (m - (1 << (K - 1))):][:4]
    for _ in range(80):
        key, w, lo, hi = (int(v) for v in rec)
        r = bool(right(key, w))
        nxt = hi if r else lo
        if w & (RIGHT_LEAF if r else LEFT_LEAF):
            return nxt
        assert 0 < nxt < deep.shape[0]
        rec = deep[nxt]
    raise AssertionError("walk does not terminate")


def _sparse_variants():
    # (the "sparse_r_*" family -- one-word nodes, pair records -- has its own walker: tests/test_sparse_r_host.py)
    return [(i, n) for i, n in enumerate(ddt.variant_names()) if n.startswith("sparse_") and not n.startswith("sparse_r_")]


@pytest.mark.parametrize("order", [0, 1])
@pytest.mark.parametrize("shape", [(19, 14, 12, 3, 600), (8, 3, 5, 1, 500), (11, 20, 30, 0, 850), (1, 1, 3, 0, 0)])
def test_packed_images_walk_to_the_oracles_leaves(shape, order):
    T, depth, F, full, pm = shape
    s = O.gen_sparse_model(T, depth, F, full, pm, 1)
    x = O.gen_tuples(3, 60, F, dist=1, missing_bits=s.params.missing_bits)
    x[::7, 0] = s.params.missing_bits  # missing values on the feature most roots test
    seen_k = set()
    tables = _rank_tables(s)
    for vid, name in _sparse_variants():
        ranked, dense = name.startswith(("sparse_q_", "sparse_qd_", "sparse_qp_")), name.startswith(("sparse_dk_", "sparse_qd_", "sparse_dm", "sparse_dp", "sparse_qp_"))
        mid = int(re.match(r"sparse_dm(\d)_", name).group(1)) if name.startswith("sparse_dm") else 0
        K = int(re.search(r"_k(\d+)_", name).group(1))
        gf = name.startswith("sparse_gf_")   # features gathered from global memory: the address field is the byte offset in the tuple's row
        pairs = name.startswith(("sparse_dp", "sparse_qp_"))
        if (ranked, dense, K, gf, mid, pairs) in seen_k:  # one variant per family and K: the packing depends on K and on the tile geometry only through the feature-row addresses
            continue
        seen_k.add((ranked, dense, K, gf, mid, pairs))
        top, deep, info = _images(s, vid, order)
        if dense and not ranked:  # the second word of heap record 0 is spare (0): no kernel reads it
            assert not top.reshape(info[2] * 8, -1)[:, 1].any(), name
        assert info[3] == K and info[2] * 8 >= T and top.size == info[2] * 8 * ((8 if dense else 12) << K) // 4
        assert info[5] == (4 if gf else 2048 if ranked else 4 * int(name.rsplit("_t", 1)[1]))  # u16 rows of 1024 tuples / fp32 rows of the tile
        assert not gf or info[4] == 0
        for r in range(x.shape[0]):
            for i in range(info[2] * 8):
                got = _walk(top, deep, info, i, x[r], int(s.params.missing_bits), tables if ranked else None, mid, pairs)
                want = O.traverse_sparse(s, x[r], i) if i < T else 0
                assert got == want, (name, order, r, i, hex(got), hex(want))
    assert len([k for k in seen_k if not k[0] and not k[1]]) >= 4 and len([k for k in seen_k if k[0]]) >= 4 and len([k for k in seen_k if k[1]]) >= 4
    assert any(k[3] for k in seen_k) and {k[4] for k in seen_k} >= {0, 1} and any(k[5] for k in seen_k)


def test_hook_rejects_what_the_loader_rejects():
    s = O.gen_sparse_model(4, 6, 5, 2, 600, 1)
    vid = _sparse_variants()[0][0]
    L = _lib.lib()
    nl = np.ascontiguousarray(s.node_lines).view(np.uint32).reshape(-1, 4).copy()
    first = np.ascontiguousarray(s.first, dtype=np.uint64)
    p = ddt.make_sparse_params(4, 6, 5)
    info = np.zeros(6, np.uint64)
    call = lambda lines, par=p, v=vid: L.ddt_debug_sparse_image(C.byref(par), lines.ctypes.data, lines.shape[0], first.ctypes.data, v, 0, None, 0, None, 0,
                                                                 info.ctypes.data)
    assert call(nl) == 0
    bad = nl.copy()
    bad[0, 1] = (int(bad[0, 1]) & 0xFFFFF800) | 9  # feature index >= num_features
    assert call(bad) < 0
    bad = nl.copy()
    internal = np.flatnonzero((bad[:, 1] >> 14) & 1 == 0)  # a node whose left child is internal
    bad[internal[0], 2] = internal[0]  # child index not after its parent
    assert call(bad) < 0
    assert call(nl, ddt.make_sparse_params(4, 3, 5)) < 0   # deeper than num_levels
    assert call(nl, p, 0) < 0                              # not a sparse kernel variant


def test_a_stream_must_be_a_tree_not_a_dag():
    """a chain whose nodes share a child passed validation and the packers then expanded 2^depth paths (bad_alloc across the
    C ABI from a 480-byte stream).  Every node but the root must be the child of exactly one earlier node."""
    L = _lib.lib()
    vid = _sparse_variants()[0][0]
    depth = 30
    lines = np.zeros((depth, 4), np.uint32)
    for n in range(depth):
        last = n == depth - 1
        lines[n] = [np.float32(0.5).view(np.uint32), (3 << 14) if last else 0, 0 if last else n + 1, 0 if last else n + 1]  # both children = node n + 1
    first = np.array([0, depth], np.uint64)
    p = ddt.make_sparse_params(1, 40, 4)
    info = np.zeros(6, np.uint64)
    rc = L.ddt_debug_sparse_image(C.byref(p), lines.ctypes.data, depth, first.ctypes.data, vid, 0, None, 0, None, 0, info.ctypes.data)
    assert rc == -1  # DDT_EINVAL, not a crash and not an exponential image
    # a node nobody points to
    s = O.gen_sparse_model(1, 6, 5, 2, 600, 1)
    nl = np.ascontiguousarray(s.node_lines).view(np.uint32).reshape(-1, 4).copy()
    internal = np.flatnonzero((nl[:, 1] >> 14) & 1 == 0)
    assert internal.size
    n = int(internal[0])
    nl[n, 1] |= 1 << 14                    # its left child becomes a leaf: the old left sub-tree is unreachable now
    nl[n, 2] = 0
    first = np.ascontiguousarray(s.first, dtype=np.uint64)
    p = ddt.make_sparse_params(1, 6, 5)
    assert L.ddt_debug_sparse_image(C.byref(p), nl.ctypes.data, nl.shape[0], first.ctypes.data, vid, 0, None, 0, None, 0, info.ctypes.data) == -1
