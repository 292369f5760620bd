"""Independent, deliberately naive pure-numpy restatement of the scoring rules (tests only).

Written separately from oracle/ddt_oracle.c (different language, different loop structure: all tuples
walk one tree level at a time) so that an agreement between the two is evidence, not a tautology.
Rules: SURVEY.md section 8(a) "normative scoring semantics"; reference rtl/DTEngine/core/DTPU.sv:579-760.
"""
import numpy as np


def unpack_model(wlines, flines, T, D, wlpt, flpt):
    """wire format -> (thr_bits [T,2^D-1], fidx, miss_right, leaf_bits [T,2^D])."""
    nint, nleaf = (1 << D) - 1, 1 << D
    w = np.asarray(wlines, np.uint32).reshape(T, wlpt * 4)
    f = np.asarray(flines, np.uint16).reshape(T, flpt * 8)
    thr = w[:, :nint].copy()
    leaf = w[:, nint:nint + nleaf].copy()
    e = f[:, :nint]
    return thr, (e & 0x7FF).astype(np.int64), ((e >> 13) & 1).astype(np.uint8), leaf


def traverse_all(thr, fidx, miss_right, leaf, x_bits, missing_bits, cmp_mode=0):
    """x_bits uint32 [n, >=F] -> leaf bits uint32 [n, T]."""
    T, nint = thr.shape
    D = int(np.log2(nint + 1))
    n = x_bits.shape[0]
    out = np.zeros((n, T), np.uint32)
    rows = np.arange(n)
    for t in range(T):
        node = np.zeros(n, np.int64)
        for _ in range(D):
            j = fidx[t, node]
            f = x_bits[rows, j]
            w = thr[t, node]
            if cmp_mode == 0:
                less = f.view(np.int32) < w.view(np.int32)
            else:
                with np.errstate(invalid="ignore"):
                    less = f.view(np.float32) < w.view(np.float32)
            right = np.where(f == np.uint32(missing_bits), miss_right[t, node].astype(bool), ~less)
            node = 2 * node + 1 + right.astype(np.int64)
        out[:, t] = leaf[t, node - nint]
    return out


def reduce_reference_order(leaf_bits_row, C):
    """fp32 sum of one tuple's leaves in the reference's adder order (one device), numpy fp32 adds."""
    l = np.asarray(leaf_bits_row, np.uint32).view(np.float32)
    T = l.size
    groups = (T + 7) // 8
    slots = (groups + C - 1) // C
    pad = np.zeros(slots * C * 8, np.float32)
    pad[:T] = l
    acc = [np.float32(0.0)] * C
    for t in range(slots):
        for c in range(C):
            g = t * C + c
            v = pad[g * 8:(g + 1) * 8]
            a0, a1, a2, a3 = v[0] + v[1], v[2] + v[3], v[4] + v[5], v[6] + v[7]
            s = (a0 + a1) + (a2 + a3)
            acc[c] = np.float32(s + acc[c])
    tot = np.float32(0.0)
    for c in range(C):
        tot = np.float32(acc[c] + tot)
    return tot


def shard_bounds(T, n_dev):
    per = (T + n_dev - 1) // n_dev
    return [(min(d * per, T), min((d + 1) * per, T)) for d in range(n_dev)]


def score_reference_order(leaf_bits, C, n_dev=1):
    """leaf_bits [n, T] -> fp32 scores, tree-sharded over n_dev devices with chain add."""
    n, T = leaf_bits.shape
    out = np.zeros(n, np.float32)
    for r in range(n):
        run = None
        for (b, e) in shard_bounds(T, n_dev):
            part = reduce_reference_order(leaf_bits[r, b:e], C) if e > b else np.float32(0.0)
            run = part if run is None else np.float32(part + run)
        out[r] = run
    return out


def pad_to_perfect(children_left, children_right, feature, threshold, value, D):
    """sklearn-style explicit tree -> perfect depth-D heap arrays (thr fp64, fidx, leaf values).

    A leaf above depth D is replaced by a dummy subtree whose leaves all carry its value
    (shallower subtrees must be padded with replicated leaves)."""
    nint, nleaf = (1 << D) - 1, 1 << D
    thr = np.zeros(nint, np.float64)
    fidx = np.zeros(nint, np.int64)
    leaf = np.zeros(nleaf, np.float64)

    def rec(sk_node, heap, depth, frozen_val):
        if depth == D:
            leaf[heap - nint] = frozen_val if sk_node < 0 else value[sk_node]
            assert sk_node < 0 or children_left[sk_node] == -1, "tree deeper than D"
            return
        if sk_node >= 0 and children_left[sk_node] != -1:
            thr[heap], fidx[heap] = threshold[sk_node], feature[sk_node]
            rec(children_left[sk_node], 2 * heap + 1, depth + 1, None)
            rec(children_right[sk_node], 2 * heap + 2, depth + 1, None)
        else:
            v = frozen_val if sk_node < 0 else value[sk_node]
            thr[heap], fidx[heap] = 0.0, 0
            rec(-1, 2 * heap + 1, depth + 1, v)
            rec(-1, 2 * heap + 2, depth + 1, v)

    rec(0, 0, 0, None)
    return thr, fidx, leaf


def sklearn_threshold_to_lt(t64):
    """sklearn goes left iff float32(x) <= t64.  Return fp32 thr with: x <= t64  <=>  x < thr."""
    t64 = np.asarray(t64, np.float64)
    t32 = t64.astype(np.float32)
    too_big = t32.astype(np.float64) > t64
    t32 = np.where(too_big, np.nextafter(t32, np.float32(-np.inf)), t32)  # largest fp32 <= t64
    return np.nextafter(t32, np.float32(np.inf)).astype(np.float32)
