#!/usr/bin/env python3
"""How the CPU baseline scales with threads on this box (what `cores` on the bench line means)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

m = O.gen_model(1000, 8, 32, 0)
x = O.gen_tuples(0, 2_000_000, 32, 0)
O.score_fast(m, x[:100_000])
for nt in (8, 16, 32, 64, 96, 128, 192, 256):
    if nt > 2 * (os.cpu_count() or 1):
        break
    t = time.time()
    O.score_fast(m, x, nthreads=nt)
    dt = time.time() - t
    print(f"{nt:4d} threads: {len(x) / dt / 1e6:7.3f} Mtuples/s  ({len(x) * 8000 / dt / nt / 1e6:6.0f} M visits/s/thread)", flush=True)
