#!/usr/bin/env python3
"""What the host link of this box gives plain pinned copies (context for the feeder's PCIe-inclusive rate)."""
import time

import torch

n = 1 << 30
h = torch.empty(n, dtype=torch.uint8).pin_memory()
d = torch.empty(n, dtype=torch.uint8, device="cuda")
s2 = torch.cuda.Stream()
for name, fn in (("H2D pinned 1 GiB", lambda: d.copy_(h, non_blocking=True)), ("D2H pinned 1 GiB", lambda: h.copy_(d, non_blocking=True))):
    fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        t = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t)
    print(f"{name}: {n / best / 1e9:.1f} GB/s")
# both directions at once (the feeder's steady state: tuples in, scores out -- 32:1 in bytes)
h2 = torch.empty(n // 32, dtype=torch.uint8).pin_memory()
d2 = torch.empty(n // 32, dtype=torch.uint8, device="cuda")
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(4):
    d.copy_(h, non_blocking=True)
    with torch.cuda.stream(s2):
        h2.copy_(d2, non_blocking=True)
torch.cuda.synchronize()
print(f"H2D 1 GiB + D2H 32 MiB concurrently: {4 * (n + n // 32) / (time.perf_counter() - t) / 1e9:.1f} GB/s total")
# chunked like the feeder: 128 MiB pieces on three streams
streams = [torch.cuda.Stream() for _ in range(3)]
torch.cuda.synchronize()
t = time.perf_counter()
step = 128 << 20
for i in range(0, n, step):
    with torch.cuda.stream(streams[(i // step) % 3]):
        d[i:i + step].copy_(h[i:i + step], non_blocking=True)
torch.cuda.synchronize()
print(f"H2D 1 GiB in 128 MiB pieces on 3 streams: {n / (time.perf_counter() - t) / 1e9:.1f} GB/s")
for step_mb, ns in ((128, 1), (32, 1), (8, 1), (512, 3), (128, 2), (32, 3)):
    streams = [torch.cuda.Stream() for _ in range(ns)]
    step = step_mb << 20
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(0, n, step):
        with torch.cuda.stream(streams[(i // step) % ns]):
            d[i:i + step].copy_(h[i:i + step], non_blocking=True)
    torch.cuda.synchronize()
    print(f"H2D 1 GiB in {step_mb} MiB pieces on {ns} stream(s): {n / (time.perf_counter() - t) / 1e9:.1f} GB/s")
