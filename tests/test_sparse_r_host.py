"""Host logic of the "sparse_r_*" family (csrc/ddt_sparse_r.hip; csrc/ddt_internal.h "32-bit ranks"), no GPU needed:

* the tables of the 32-bit rank pre-pass (`ddt_sparse_host.cpp pack_rank32_tables`: key blocks + the directory of their last keys + bucket
  starts) come back through `ddt_debug_rank32_tables` and rank32_kernel's search is replayed on them in numpy against a plain sorted-table
  count -- tables from 0 to 131,070 keys = the most a 17-bit rank holds (block sizes 4 and 8), keys at both ends of the int32 range, values on, between and beyond the keys;
* the images (`sparse_pack_host_r`: one-word nodes in the top heap, pair / LEAF records below) come back through `ddt_debug_sparse_image`
  and are walked the way score_sparse_r_kernel walks them; every (tuple, tree) must end on the leaf the oracle's walk of the wire format
  ends on (DTPU.sv:579-720 semantics), EMPTY slots on +0, within the number of record hops the packer reports.
"""
import ctypes as C

import numpy as np
import pytest

from oracle import oracle as O
import ddt
from ddt import _lib

LEFT_LEAF, RIGHT_LEAF, MISS_RIGHT, FEAT_SHIFT, FEAT, RANK_SHIFT = 0x80, 0x40, 0x20, 8, 0x7F, 15  # csrc/ddt_internal.h kSr*
MISSING = 0xFFFFFFFF
BUCKETS = 4096


def _tables(keys_per_word):
    L = _lib.lib()
    L.ddt_debug_rank32_tables.restype = C.c_int
    W = len(keys_per_word)
    keys = np.concatenate([np.asarray(k, np.int32) for k in keys_per_word] + [np.zeros(0, np.int32)]).view(np.uint32)
    counts = np.asarray([len(k) for k in keys_per_word], np.uint32)
    info = np.zeros(4, np.uint64)
    vp = C.c_void_p
    L.ddt_debug_rank32_tables.argtypes = [vp, vp, C.c_uint32, vp, C.c_size_t, vp, vp, vp, C.c_size_t, vp]
    rc = L.ddt_debug_rank32_tables(keys.ctypes.data, counts.ctypes.data, W, None, 0, None, None, None, 0, info.ctypes.data)
    assert rc == 0, rc
    Kpad, bl, tabw = int(info[0]), int(info[1]), int(info[2])
    d = np.zeros(W * Kpad, np.uint32)
    par = np.zeros(W * 8, np.uint32)
    st = np.zeros(W * BUCKETS, np.uint16)
    tab = np.zeros(tabw, np.uint32)
    rc = L.ddt_debug_rank32_tables(keys.ctypes.data, counts.ctypes.data, W, d.ctypes.data, d.size, par.ctypes.data, st.ctypes.data, tab.ctypes.data,
                                   tab.size, info.ctypes.data)
    assert rc == 0, rc
    return d.reshape(W, Kpad).view(np.int32), par.reshape(W, 8), st.reshape(W, BUCKETS), tab.view(np.int32), Kpad, bl


def _rank32(d, par, st, tab, Kpad, bl, j, x):
    """rank32_kernel's search for the int32 key values `x` on feature j, vectorised"""
    Kd, lo, _, shift, P, koff, K, hi_real = (int(v) for v in par[j])
    lo_i, hi_i = np.int32(np.uint32(lo)), np.int32(np.uint32(hi_real))
    xu = x.view(np.uint32).astype(np.uint64)
    b = ((xu - lo) & 0xFFFFFFFF) >> shift
    b = np.minimum(b, BUCKETS - 1)
    b = np.where(x < lo_i, 0, b).astype(np.int64)
    pos = st[j][b].astype(np.int64)
    step = P >> 1
    while step >= 1:
        probe = np.minimum(pos + step - 1, Kpad - 1)
        pos = pos + np.where(d[j][probe] <= x, step, 0)
        step >>= 1
    pos = np.minimum(pos, Kd)
    B = 1 << bl
    cnt = np.zeros_like(pos)
    for i in range(B):
        cnt += tab[koff + pos * B + i] <= x
    r = np.minimum(pos * B + cnt, K)
    return np.where(x >= hi_i, K, r)


def test_rank32_tables_replay_the_kernels_search():
    rng = np.random.default_rng(11)
    imin, imax = np.iinfo(np.int32).min, np.iinfo(np.int32).max
    float_keys = np.unique(rng.random(200_000, dtype=np.float32))[:131_070].view(np.int32)        # the largest table: > 4 * 32767 keys -> blocks of 8
    words = [
        np.zeros(0, np.int32),                                                                  # unused feature: every value ranks 0
        np.asarray([5], np.int32),
        np.asarray([-7, 0, 9], np.int32),
        np.asarray([imin, -1, 0, 1, imax], np.int32),                                           # both ends of the range, one block + one key
        np.unique(rng.integers(-2_000_000, 2_000_000, 40_000).astype(np.int32)),
        np.unique(rng.integers(imin, imax, 1000).astype(np.int32)),
        np.unique((rng.random(3000, dtype=np.float32) * 0.1 + 0.5).view(np.int32)),             # a narrow key range (one float binade)
        float_keys,
    ]
    d, par, st, tab, Kpad, bl = _tables(words)
    assert bl == 3 and Kpad & (Kpad - 1) == 0 and Kpad > (float_keys.size + 7) // 8
    for j, k in enumerate(words):
        assert int(par[j][6]) == k.size and int(par[j][0]) == (k.size + (1 << bl) - 1) >> bl
        xs = [rng.integers(imin, imax, 20_000).astype(np.int32), np.asarray([imin, imin + 1, -1, 0, 1, imax - 1, imax], np.int32)]
        if k.size:
            pick = k[rng.integers(0, k.size, 20_000)]
            xs += [pick, pick - 1, pick + 1, k[:64], k[-64:]]                                   # on a key, right below, right above; the table's ends
        x = np.concatenate(xs).astype(np.int32)
        want = np.searchsorted(k, x, side="right")
        got = _rank32(d, par, st, tab, Kpad, bl, j, x)
        bad = np.nonzero(got != want)[0]
        assert bad.size == 0, (j, k.size, x[bad[:5]], got[bad[:5]], want[bad[:5]])
    # tables that fit blocks of 4
    d, par, st, tab, Kpad, bl = _tables(words[:7] + [np.zeros(0, np.int32)])
    assert bl == 2
    for j, k in enumerate(words[:7]):
        x = np.concatenate([rng.integers(imin, imax, 5000).astype(np.int32), k, k - 1, k + 1]).astype(np.int32) if k.size else rng.integers(imin, imax, 100).astype(np.int32)
        assert np.array_equal(_rank32(d, par, st, tab, Kpad, bl, j, x), np.searchsorted(k, x, side="right")), j


def _images(s, variant):
    L = _lib.lib()
    nl = np.ascontiguousarray(s.node_lines).view(np.uint32).reshape(-1, 4)
    first = np.ascontiguousarray(s.first, dtype=np.uint64)
    q = s.params
    p = ddt.make_sparse_params(q.num_trees, q.num_levels, q.num_features, q.missing_bits, q.cmp_mode, q.clusters_per_tuple, 0)
    info = np.zeros(6, np.uint64)
    rc = L.ddt_debug_sparse_image(C.byref(p), nl.ctypes.data, nl.shape[0], first.ctypes.data, variant, 0, None, 0, None, 0, info.ctypes.data)
    assert rc == 0, rc
    top, deep = np.zeros(int(info[0]), np.uint32), np.zeros(int(info[1]), np.uint32)
    rc = L.ddt_debug_sparse_image(C.byref(p), nl.ctypes.data, nl.shape[0], first.ctypes.data, variant, 0, top.ctypes.data, top.size, deep.ctypes.data,
                                  deep.size, info.ctypes.data)
    assert rc == 0, rc
    return top, deep, [int(v) for v in info]


def _walk(top, deep, K, slot, xr, slow):
    """score_sparse_r_kernel's walk of one tree slot: xr = the tuple's rank words (rank << 15 | 0x7FFF, or MISSING)"""
    t = top[slot << K: (slot + 1) << K]

    def right(rec):
        f = int(xr[(rec >> FEAT_SHIFT) & FEAT])
        if slow and f == MISSING:
            return int((rec & MISS_RIGHT) != 0)
        return int(f >= rec)

    m = 1
    for _ in range(K):
        m = 2 * m + right(int(t[m]))
    byte = (int(t[0]) + 16 * m) & 0xFFFFFFFF
    for hop in range(1, 80):
        assert byte % 16 == 0 and byte // 4 + 3 < deep.size
        n, a, b, ptr = (int(v) for v in deep[byte // 4: byte // 4 + 4])
        r0 = right(n)
        cw = b if r0 else a
        if n & (RIGHT_LEAF if r0 else LEFT_LEAF):
            return cw, hop
        byte = (ptr + 32 * r0 + 16 * right(cw)) & 0xFFFFFFFF
    raise AssertionError("walk does not terminate")


@pytest.mark.parametrize("shape", [(19, 14, 12, 3, 600), (8, 3, 5, 1, 500), (11, 20, 30, 0, 850), (1, 1, 3, 0, 0), (9, 16, 64, 10, 700)])
def test_packed_r32_images_walk_to_the_oracles_leaves(shape):
    T, depth, F, full, pm = shape
    s = O.gen_sparse_model(T, depth, F, full, pm, 1)
    x = O.gen_tuples(3, 40, F, dist=1, missing_bits=s.params.missing_bits)
    x[::7, 0] = s.params.missing_bits
    nl = np.ascontiguousarray(s.node_lines).view(np.uint32).reshape(-1, 4)
    tables = [np.unique(nl[(nl[:, 1] & 0x7FF) == j, 0].view(np.int32)) for j in range(x.shape[1])]
    miss = x == np.uint32(s.params.missing_bits)
    xr = np.stack([np.searchsorted(tables[j], x[:, j].view(np.int32), side="right").astype(np.uint64) for j in range(x.shape[1])], axis=1)
    xr = ((xr << RANK_SHIFT) | ((1 << RANK_SHIFT) - 1)).astype(np.uint32)
    xr[miss] = MISSING
    seen = set()
    for vid, name in enumerate(ddt.variant_names()):
        if not name.startswith("sparse_r_"):
            continue
        K = int(name.split("_k")[1].split("_")[0])
        U = int(name.split("_u")[1].split("_")[0])
        if (K, U) in seen:  # the packing depends on K and on the PU groups per pass (padding to whole passes)
            continue
        seen.add((K, U))
        top, deep, info = _images(s, vid)
        groups, rounds = info[2], info[3] >> 32
        assert info[3] & 0xFFFFFFFF == K and groups * 8 >= T and groups % (U // 8) == 0 and top.size == groups * 8 << K
        deepest = 0
        for r in range(x.shape[0]):
            for i in range(groups * 8):
                got, hops = _walk(top, deep, K, i, xr[r], bool(miss[r].any()))
                want = O.traverse_sparse(s, x[r], i) if i < T else 0
                assert got == want, (name, r, i, hex(got), hex(want))
                assert hops <= rounds, (name, hops, rounds)
                deepest = max(deepest, hops)
        # the packer's round count is tight on the forest (not on 40 tuples): every level below the dense block costs half a hop, a leaf two
        # levels below a pair record one more
        assert 1 <= rounds <= max(1, (depth - K + 1) // 2 + 1), (name, rounds)
    assert seen
