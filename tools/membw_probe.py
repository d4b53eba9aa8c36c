#!/usr/bin/env python3
"""Quick on-box ceilings for the memory system with torch elementwise kernels (float4-vectorised):
pure write (fill), copy (1R+1W), add (2R+1W), at an HBM-sized and an Infinity-Cache-sized working set."""
import torch

dev = torch.device("cuda")


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3


for mb in (16, 64, 256, 1024):
    n = mb * (1 << 20) // 8
    a = torch.empty(n, dtype=torch.float64, device=dev).normal_()
    b = torch.empty_like(a).normal_()
    c = torch.empty_like(a)
    t = timeit(lambda: c.fill_(1.5))
    print(f"{mb:5d} MiB  fill  (1W)    {mb*1.048576e6/t/1e9:8.0f} GB/s   {t*1e6:8.1f} us")
    t = timeit(lambda: c.copy_(a))
    print(f"{mb:5d} MiB  copy  (1R+1W) {2*mb*1.048576e6/t/1e9:8.0f} GB/s   {t*1e6:8.1f} us")
    t = timeit(lambda: torch.add(a, b, out=c))
    print(f"{mb:5d} MiB  add   (2R+1W) {3*mb*1.048576e6/t/1e9:8.0f} GB/s   {t*1e6:8.1f} us")
    t = timeit(lambda: a.sum())
    print(f"{mb:5d} MiB  sum   (1R)    {mb*1.048576e6/t/1e9:8.0f} GB/s   {t*1e6:8.1f} us")
