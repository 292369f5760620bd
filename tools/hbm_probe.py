#!/usr/bin/env python3
"""What does this box's HBM deliver to simple kernels?  (context for the stream kernel's and the pre-pass's GB/s)"""
import torch
x = torch.empty(3_200_000_000, dtype=torch.float32, device="cuda").normal_()  # 12.8 GB
y = torch.empty_like(x)
def t(f, n=5):
    f(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(n):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); f(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best
gb = x.numel() * 4 / 1e9
ms = t(lambda: x.sum()); print(f"read-only   sum(fp32 {gb:.1f} GB): {ms:7.3f} ms  {gb / ms * 1e3 / 1e3:6.2f} TB/s")
ms = t(lambda: x.view(torch.int32).max()); print(f"read-only   max(int32):          {ms:7.3f} ms  {gb / ms:6.2f} TB/s")
ms = t(lambda: y.copy_(x)); print(f"copy r+w    {2 * gb:.1f} GB moved:        {ms:7.3f} ms  {2 * gb / ms:6.2f} TB/s")
ms = t(lambda: y.fill_(1.0)); print(f"write-only  fill:                {ms:7.3f} ms  {gb / ms:6.2f} TB/s")
ms = t(lambda: torch.add(x, 1.0, out=y)); print(f"r+w add     {2 * gb:.1f} GB moved:        {ms:7.3f} ms  {2 * gb / ms:6.2f} TB/s")
