#!/usr/bin/env python3
"""Reference points for HBM-bound kernels on this GPU: fill, copy, and a strided 2-D copy."""
import torch

def t(fn, reps=50):
    for _ in range(5):
        fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3

for mb in (42, 85, 340, 1360):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device="cuda")
    b = torch.randn(n, device="cuda")
    us = t(lambda: a.fill_(1.0))
    print("fill  %5d MB  %7.1f us  %5.2f TB/s" % (mb, us, n * 4 / us / 1e6))
    us = t(lambda: a.copy_(b))
    print("copy  %5d MB  %7.1f us  %5.2f TB/s (read+write)" % (mb, us, 2 * n * 4 / us / 1e6))
    us = t(lambda: torch.relu_(a))
    print("relu_ %5d MB  %7.1f us  %5.2f TB/s (read+write)" % (mb, us, 2 * n * 4 / us / 1e6))
