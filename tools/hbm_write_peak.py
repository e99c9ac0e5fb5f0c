#!/usr/bin/env python3
"""Practical HBM write ceiling on this GPU: torch fill_ / zero_ / copy_ over a 9 GB buffer (the size one encode launch
writes at 65,536 tables), timed with CUDA events.  Context for roofline.frac (spec peak 8 TB/s is never reached by writes)."""
import torch

n = 9_170_000_000 // 4
x = torch.empty(n, dtype=torch.float32, device="cuda")
y = torch.empty(n, dtype=torch.float32, device="cuda")


def timeit(f, reps=10):
    for _ in range(3):
        f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        f()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for name, f, bytes_ in (("fill_(1.0)", lambda: x.fill_(1.0), n * 4), ("zero_()", lambda: x.zero_(), n * 4),
                        ("copy_ (r+w)", lambda: y.copy_(x), 2 * n * 4)):
    ms = timeit(f)
    print(f"{name:14s} {ms:8.3f} ms  {bytes_ / ms / 1e6:8.1f} GB/s")
