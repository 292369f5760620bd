"""Malformed and random model streams through the host-side validators and packers (no GPU needed): the library must
answer with a DDT_E* code or a packed image, never crash, and an accepted sparse forest must pack for every sparse
variant.  (ddt_debug_model_image / ddt_debug_sparse_image run exactly the validation + packing of ddt_load_model /
ddt_load_model_sparse; the seeded cases below are a trimmed copy of the fuzz loop that was run for 1,600 cases.)"""
import ctypes as C

import numpy as np
import pytest

import ddt
from oracle import oracle as O


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_random_and_corrupted_streams_never_crash_the_host_half(seed):
    L = ddt.lib()
    rng = np.random.default_rng(seed)
    info, sinfo = (C.c_uint64 * 12)(), (C.c_uint64 * 6)()
    names = ddt.variant_names()
    sparse_ids = [i for i, n in enumerate(names) if n.startswith("sparse")]
    seen = set()
    for _ in range(60):
        T, D, F = int(rng.integers(1, 40)), int(rng.integers(1, 11)), int(rng.choice([1, 3, 16, 28, 32, 33, 64, 200]))
        p = ddt.make_params(T, D, F, cmp_mode=int(rng.integers(0, 2)))
        nint = (1 << D) - 1
        w = rng.integers(0, 1 << 32, T * p.weights_lines_per_tree * 4, dtype=np.uint64).astype(np.uint32)
        if rng.random() < 0.7:  # leaves inside the exact domain of the reference adder
            w.reshape(T, -1)[:, nint:nint + (1 << D)] = (rng.random((T, 1 << D)).astype(np.float32) - 0.5).view(np.uint32)
        f = (rng.integers(0, F, T * p.findex_lines_per_tree * 8, dtype=np.uint64) |
             (rng.integers(0, 2, T * p.findex_lines_per_tree * 8, dtype=np.uint64) << 13)).astype(np.uint16)
        if rng.random() < 0.15:
            f[int(rng.integers(0, f.size))] |= np.uint16(1 << 14)      # early-leaf flag: refused unless it sits in padding
        if rng.random() < 0.15:
            f[int(rng.integers(0, f.size))] = np.uint16(F)            # feature index out of range
        v = int(rng.integers(-1, len(names)))
        rc = L.ddt_debug_model_image(C.byref(p), w.ctypes.data, w.size // 4, f.ctypes.data, f.size // 8, v, None, None, 0, None, 0, C.byref(info))
        assert rc in (0, -1, -5), rc
        seen.add(("model", rc))
        if rc == 0:
            img, slow, tab = np.zeros(info[0], np.uint32), np.zeros(info[0], np.uint32), np.zeros(max(info[11], 1), np.uint32)
            assert L.ddt_debug_model_image(C.byref(p), w.ctypes.data, w.size // 4, f.ctypes.data, f.size // 8, int(info[10]), img.ctypes.data,
                                           slow.ctypes.data, img.size, tab.ctypes.data, tab.size, C.byref(info)) == 0
        # a valid synthetic sparse forest, then one corruption
        Ts, Dm, Fs = int(rng.integers(1, 16)), int(rng.integers(1, 13)), int(rng.choice([1, 5, 20, 64]))
        s = O.gen_sparse_model(Ts, Dm, Fs, min(Dm, int(rng.integers(0, 5))), int(rng.integers(0, 1000)), int(rng.integers(0, 2)))
        lines, first = s.node_lines.copy().reshape(-1, 4), s.first.copy()
        mode = int(rng.integers(0, 6))
        if lines.shape[0] and mode:
            k = int(rng.integers(0, lines.shape[0]))
            if mode == 1:
                lines[k, 2] = rng.integers(0, 1 << 32)                   # child word: junk index (or, flagged, any leaf bits)
            elif mode == 2:
                lines[k, 1] = rng.integers(0, 1 << 16)                   # entry word
            elif mode == 3:
                lines[k, 3] = k                                          # a node that is its own right child
            elif mode == 4:
                first[int(rng.integers(0, first.size))] = rng.integers(0, lines.shape[0] + 5)
            else:
                lines[k, 1] &= 0xFFFF3FFF                                # both children internal, whatever the words say
        ps = ddt.make_sparse_params(Ts, Dm, Fs)
        rcs = [L.ddt_debug_sparse_image(C.byref(ps), lines.ctypes.data, lines.shape[0], first.ctypes.data, vid, 0, None, 0, None, 0, C.byref(sinfo))
               for vid in sparse_ids]
        assert all(r in (0, -1, -5) for r in rcs), rcs
        assert len({r == 0 for r in rcs}) == 1 or -5 in rcs              # accepted by validation => packs for every variant that fits
        if mode == 0:
            assert 0 in rcs
        seen.add(("sparse", mode, rcs[0]))
    assert ("model", 0) in seen or ("model", -5) in seen
    assert any(k[0] == "sparse" and k[2] == -1 for k in seen) and any(k[0] == "sparse" and k[2] == 0 for k in seen)


def test_inner_entry_of_tree_first_line_past_the_stream_is_refused():
    """Found by the AddressSanitizer run of the fuzz loop on the model build (tests/mock_hip/README.md): tree_first_line[0] = 0 and
    tree_first_line[T] <= n_lines were checked, an INNER entry pointing past the stream was not -- the validator read the lines of a
    tree that was not there before it reached the entry that is out of order."""
    L = ddt.lib()
    s = O.gen_sparse_model(3, 6, 8, 2, 600, 1)
    lines = np.ascontiguousarray(s.node_lines, np.uint32).reshape(-1, 4)
    first = np.ascontiguousarray(s.first, np.uint64).copy()
    n = lines.shape[0]
    first[1], first[2] = n + 50, n + 60                      # strictly increasing up to there, then back inside the stream
    sinfo = (C.c_uint64 * 6)()
    vid = [i for i, nm in enumerate(ddt.variant_names()) if nm.startswith("sparse")][0]
    p = ddt.make_sparse_params(3, 6, 8)
    assert L.ddt_debug_sparse_image(C.byref(p), lines.ctypes.data, n, first.ctypes.data, vid, 0, None, 0, None, 0, C.byref(sinfo)) == -1
