"""Host logic of the LDS-resident rank pre-pass, no GPU needed: the images `ddt_image.cpp build_prepass_group` packs (linear since round 6; until then skewed
key tables + pads, segmented bucket index, parameter blocks) are built through the test hook
`ddt_debug_prepass_image` and the kernel's search (`ddt_prepass.hip rank_line`: clamp, segment lookup, bucket start, log2 P
probes without an end test) is replayed on them in numpy against a plain count of the keys <= x."""
import ctypes as C

import numpy as np
import pytest

from ddt import _lib

LDS_BYTES = 160 * 1024


def _build(tables, groups):
    """tables: list of W sorted int32 arrays -> (plan, image words) or (None, None) when the tables do not fit"""
    L = _lib.lib()
    W = len(tables)
    keys = np.concatenate([t.astype(np.int32) for t in tables] + [np.zeros(0, np.int32)]).view(np.uint32)
    keys = np.ascontiguousarray(keys if keys.size else np.zeros(1, np.uint32))
    counts = np.array([t.size for t in tables], np.uint32)
    plan = np.zeros(42, np.uint32)
    words = L.ddt_debug_prepass_image(keys.ctypes.data, counts.ctypes.data, W, groups, None, 0, plan.ctypes.data)
    assert words >= 0, words
    if plan[0] == 0:
        return None, None
    img = np.zeros(words, np.uint32)
    assert L.ddt_debug_prepass_image(keys.ctypes.data, counts.ctypes.data, W, groups, img.ctypes.data, words, plan.ctypes.data) == words
    return plan, img


def _replay(img, par_off, P, j, x):
    """rank_line for feature j of a group image, vectorised over the int32 values x"""
    par = img[par_off // 4 + 8 * j: par_off // 4 + 8 * j + 8]
    K, lo, span, tab_off, starts_off, seg_off, seg_shift = (int(v) for v in par[:7])
    lo_s = int(np.uint32(lo).view(np.int32))
    xs = x.astype(np.int64)
    d = (xs - lo_s) & 0xFFFFFFFF
    d = np.where(xs < lo_s, 0, d)
    d = np.minimum(d, span)
    sg = img[seg_off // 4 + (d >> seg_shift)].astype(np.int64)
    bk = (sg & 0xFFFF) + ((d & ((1 << seg_shift) - 1)) >> (sg >> 16))
    pos = img.view(np.uint16)[starts_off // 2 + bk].astype(np.int64)
    tab = img[tab_off // 4:].view(np.int32)
    step = P >> 1
    while step >= 1:
        probe = pos + step - 1
        assert probe.max() < K + P  # inside the padded table
        pos = np.where(tab[probe] <= xs, pos + step, pos)   # (round 6: linear tables)
        step >>= 1
    return np.minimum(pos, K)


def _table(rng, kind, k):
    if kind == 0:
        v = rng.random(k).astype(np.float32)
    elif kind == 1:
        v = np.exp2(rng.uniform(-30, 20, k)).astype(np.float32)
    elif kind == 2:
        v = rng.integers(-50, 400, k).astype(np.float32)
    elif kind == 3:
        v = np.full(k, 0.75, np.float32)
    elif kind == 4:
        v = (-np.exp2(rng.uniform(-3, 3, k))).astype(np.float32)
    elif kind == 5:
        v = rng.choice(np.array([-3.0e38, 3.0e38, 1e-30, -1e-30, 0.0, 1.0], np.float32), k)
    elif kind == 6:
        v = np.where(rng.random(k) < 0.5, 0.25 + rng.integers(0, 4000, k) * 2.0 ** -24, 1000.0 + rng.integers(0, 4000, k) * 2.0 ** -12).astype(np.float32)
    else:  # raw bit patterns over the whole int32 range (cmp_mode 0 compares bits)
        return np.unique(rng.integers(-2 ** 31, 2 ** 31, k).astype(np.int32))
    return np.unique(v.view(np.int32))  # sorted as signed int32 = the comparator's order (cmp_mode 0)


@pytest.mark.parametrize("seed", range(10))
def test_replayed_search_counts_the_keys_on_every_group_count(seed):
    rng = np.random.default_rng(4242 + seed)
    W = int(rng.choice([4, 8, 20, 32]))
    big = seed % 3 == 0
    tables = []
    for w in range(W):
        kind = int(rng.integers(0, 8))
        k = int(rng.integers(0, 9000 if big else 1400))
        tables.append(_table(rng, kind, k) if k else np.zeros(0, np.int32))
    built = 0
    for groups in (0, 1, 2, 4, 8):
        plan, img = _build(tables, groups)
        if plan is None:
            assert groups in (1, 2, 4) or max(t.size for t in tables) > 9000  # 8 groups take ~9.5 k keys per feature
            continue
        built += 1
        G, lines = int(plan[0]), int(plan[1])
        assert lines in (1, 2, 4, 8) and (groups == 0 or lines == 8 // groups) and G == -(-(W // 4) // lines)  # narrow tuples: fewer groups
        for g in range(G):
            off, nbytes, par_off, P, line_lo = (int(v) for v in plan[2 + 5 * g: 7 + 5 * g])
            assert nbytes <= LDS_BYTES and off % 16 == 0 and nbytes % 16 == 0 and P >= 2 and P & (P - 1) == 0
            gimg = img[off // 4: (off + nbytes) // 4]
            for j in range(4 * lines):
                f = 4 * line_lo + j
                if f >= W:
                    break
                t = tables[f]
                pick = t[rng.integers(0, t.size, 3000)].astype(np.int64) + rng.integers(-2, 3, 3000) if t.size else np.zeros(0, np.int64)
                x = np.concatenate([pick, rng.integers(-2 ** 31, 2 ** 31, 2000), [-2 ** 31, 2 ** 31 - 1, 0, -1, 1]])
                x = np.clip(x, -2 ** 31, 2 ** 31 - 1).astype(np.int32)
                got = _replay(gimg, par_off, P, j, x)
                want = np.searchsorted(t, x, side="right")
                bad = np.flatnonzero(got != want)
                assert bad.size == 0, f"groups {groups} feature {f}: x={x[bad[:3]]} got {got[bad[:3]]} want {want[bad[:3]]}"
    assert built >= 1


def test_fullest_bucket_stays_below_p_and_degenerate_tables():
    """P is a power of two above the fullest bucket (that is what lets the search skip the end test); one-key, two-key and empty
    tables, and a table as long as 16-bit ranks allow (it fits only in 8 groups, or not at all)."""
    rng = np.random.default_rng(7)
    tables = [np.array([5], np.int32), np.array([-7, 9], np.int32), np.zeros(0, np.int32), _table(rng, 0, 3000)]
    plan, img = _build(tables, 0)
    assert plan is not None
    off, nbytes, par_off, P, line_lo = (int(v) for v in plan[2:7])
    g = img[off // 4: (off + nbytes) // 4]
    for j, t in enumerate(tables):
        x = np.concatenate([t.astype(np.int64) + d for d in (-1, 0, 1)] + [np.array([-2 ** 31, 2 ** 31 - 1])]).clip(-2 ** 31, 2 ** 31 - 1).astype(np.int32)
        assert np.array_equal(_replay(g, par_off, P, j, x), np.searchsorted(t, x, side="right"))
        # bucket occupancy from the starts array itself: consecutive differences stay below P
        par = g[par_off // 4 + 8 * j: par_off // 4 + 8 * j + 8]
        K, starts_off, seg_off, seg_shift = int(par[0]), int(par[4]), int(par[5]), int(par[6])
        segs = g[seg_off // 4: seg_off // 4 + 32].astype(np.int64)
        nseg = (int(par[2]) >> seg_shift) + 1
        nb = int((segs[nseg - 1] & 0xFFFF) + (1 << (seg_shift - (segs[nseg - 1] >> 16))))
        st = g.view(np.uint16)[starts_off // 2: starts_off // 2 + nb].astype(np.int64)
        occ = np.diff(np.concatenate([st, [K]]))
        assert occ.min() >= 0 and occ.max() < P and occ.sum() == K
    long_table = [np.arange(0, 32767 * 64, 64, dtype=np.int32)] + [np.zeros(0, np.int32)] * 3
    plan, _ = _build(long_table, 0)
    assert plan is not None and int(plan[0]) >= 1       # 32767 keys x 4 B = 128 KiB: fits one CU's LDS on its own
    L = _lib.lib()
    counts = np.array([40000, 0, 0, 0], np.uint32)
    keys = np.arange(40000, dtype=np.uint32)
    assert L.ddt_debug_prepass_image(keys.ctypes.data, counts.ctypes.data, 4, 0, None, 0, np.zeros(42, np.uint32).ctypes.data) < 0  # > 16-bit ranks


def _tables_of_model(T, D, F, shard=None):
    """sorted distinct threshold bits per tuple word of the synthetic benchmark model (cmp_mode 0: signed-int order of the bits)"""
    import ddt
    w, f = ddt.synth_model(T, D, F)
    nint = (1 << D) - 1
    wt = np.ascontiguousarray(w).view(np.uint32).reshape(T, -1)[:, :nint]
    ft = np.ascontiguousarray(f).view(np.uint16).reshape(T, -1)[:, :nint] & 0x7FF
    if shard is not None:
        per = -(-T // shard[1])
        wt, ft = wt[shard[0] * per:(shard[0] + 1) * per], ft[shard[0] * per:(shard[0] + 1) * per]
    W = 4 * ((F + 3) // 4)
    return [np.unique(wt[ft == j].view(np.int32)) for j in range(W)]


def test_plan_of_the_headline_model_and_its_shards():
    """What the engine picks for BASELINE config 3 and its tree shards (profiles/archive/r02_prepass_ab_grid.log): 1000 trees
    -> 8 feature groups (one tuple line each), 500 -> 4, 250 -> 4 or 2, the 125-tree shard of an 8-GPU job -> 2 groups with P = 8."""
    expect = {None: (8, 16), (0, 2): (4, 16), (1, 4): (None, None), (3, 8): (2, 8)}
    for shard, (groups, P) in expect.items():
        plan, img = _build(_tables_of_model(1000, 8, 32, shard), 0)
        assert plan is not None
        G, lines = int(plan[0]), int(plan[1])
        Ps = [int(plan[2 + 5 * g + 3]) for g in range(G)]
        if groups is None:
            assert G in (2, 4) and max(Ps) <= 16
        else:
            assert (G, max(Ps)) == (groups, P), (shard, G, Ps)
        assert G * lines == 8 and all(int(plan[2 + 5 * g + 1]) <= LDS_BYTES for g in range(G))
