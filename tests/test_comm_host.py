"""Host-side arithmetic of the multi-GPU pipeline (no GPU needed): the chunk schedule of a sharded call
(csrc/ddt_comm.cpp chunk_schedule, exported as ddt_comm_chunk_schedule) and the tree split (ddt_shard_range)."""
import ctypes as C

import numpy as np
import pytest

import ddt


def _sched(n, rows, taper, min_rows=1 << 20):
    L = ddt.lib()
    k = L.ddt_comm_chunk_schedule(n, rows, taper, min_rows, None, 0)
    assert k >= 0
    a = np.zeros(max(k, 1), np.uint64)
    assert L.ddt_comm_chunk_schedule(n, rows, taper, min_rows, a.ctypes.data, k) == k
    return [int(x) for x in a[:k]]


@pytest.mark.parametrize("n,rows", [(100_000_000, 12_500_000), (10_000_000, 12_500_000), (99_999_999, 12_500_000), (5000, 1024), (7, 7),
                                    (12_500_001, 12_500_000), (3_000_000, 1 << 20), (1, 5)])
def test_schedule_covers_the_rows_once_with_pieces_no_longer_than_a_chunk(n, rows):
    for taper, mn in ((0, 1 << 20), (1, 1 << 20), (1, 64)):
        s = _sched(n, rows, taper, mn)
        assert sum(s) == n and all(0 < m <= rows for m in s)
        if not taper:
            assert s == [rows] * (n // rows) + ([n % rows] if n % rows else [])
        else:
            plain = _sched(n, rows, 0)
            assert s[:len(plain) - 1] == plain[:-1]                      # only the final stretch changes
            tail = s[len(plain) - 1:]
            assert sum(tail) == plain[-1] and tail == sorted(tail, reverse=True)
            assert all(m % 1024 == 0 for m in tail[:-1] if m >= 1024)    # whole tiles except the very last piece


def test_headline_job_ends_on_a_quarter_chunk():
    s = _sched(100_000_000, 12_500_000, 1)
    assert len(s) == 10 and s[:7] == [12_500_000] * 7 and s[7:] == [6_250_496, 3_125_248, 3_124_256]
    assert _sched(100_000_000, 12_500_000, 0) == [12_500_000] * 8
    assert _sched(0, 5, 1) == []
    assert ddt.lib().ddt_comm_chunk_schedule(5, 0, 1, 1, None, 0) < 0


@pytest.mark.parametrize("T,G", [(1000, 8), (1000, 3), (9, 4), (5, 5), (1, 1), (37, 8)])
def test_shard_range_is_the_contiguous_ceil_split(T, G):
    covered = []
    for g in range(G):
        b, e = C.c_uint32(), C.c_uint32()
        assert ddt.lib().ddt_shard_range(T, g, G, C.byref(b), C.byref(e)) == 0
        assert e.value - b.value <= -(-T // G) and (b.value, e.value) == ddt.shard_bounds(T, G)[g]
        covered += list(range(b.value, e.value))
    assert covered == list(range(T))
    b, e = C.c_uint32(), C.c_uint32()
    assert ddt.lib().ddt_shard_range(T, G, G, C.byref(b), C.byref(e)) < 0 and ddt.lib().ddt_shard_range(T, 0, 0, C.byref(b), C.byref(e)) < 0
