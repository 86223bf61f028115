#!/usr/bin/env python3
"""tools/h2d_bench.py: what the host link gives -- pinned host -> HBM copies of the sizes cq_query_packed uses
(2 M-read chunks of 28-byte rows = 56 MB, and one large copy), alone on the GPU.  The PCIe-fed classify rate
(bench.py's host_fed leg) cannot exceed bytes-per-read / this."""
import time
import torch

assert torch.cuda.is_available()
for mb in (56, 256, 1400):
    n = mb * 1000 * 1000
    h = torch.empty(n, dtype=torch.uint8).pin_memory()
    d = torch.empty(n, dtype=torch.uint8, device="cuda")
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            d.copy_(h, non_blocking=True)
        s.synchronize()
        reps = max(4, 4000 // mb)
        t0 = time.perf_counter()
        for _ in range(reps):
            d.copy_(h, non_blocking=True)
        s.synchronize()
        dt = time.perf_counter() - t0
    print(f"pinned -> HBM, {mb:5d} MB per copy, {reps} copies back to back: {n * reps / dt / 1e9:6.2f} GB/s", flush=True)
