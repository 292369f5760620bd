#!/usr/bin/env python3
"""PCIe-inclusive rate of ddt_score (host buffers through the pinned double-buffered feeder)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "distributed-decisiontrees_amd"))
import numpy as np  # noqa: E402

import ddt  # noqa: E402

T, D, F, N = 1000, 8, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
e = ddt.Engine(0)
w, f = ddt.synth_model(T, D, F)
e.load_model(ddt.make_params(T, D, F), w, f)
x = ddt.synth_tuples_host(0, N, F)
for threads in (1, 4, 8, 16):
    e.set_option("feeder_threads", threads)
    for rows in (1 << 18, 1 << 20, 1 << 22):
        e.set_option("feeder_rows", rows)
        e.score(x[: rows * 2])
        t0 = time.perf_counter()
        out = e.score(x)
        dt = time.perf_counter() - t0
        print(f"feeder_threads {threads:>2} feeder_rows {rows:>8}: {N / dt / 1e6:8.1f} Mtuples/s  ({N * 132 / dt / 1e9:6.2f} GB/s over PCIe both ways, {dt * 1e3:.1f} ms for {N} tuples)", flush=True)
e.close()
