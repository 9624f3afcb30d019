#!/usr/bin/env python3
"""Time the batched MST read-out at several graph sizes (65 536 random triangular rows of 10 / 32 / 48 / 64 nodes,
8 192 full matrices of each size): mst_sizes.py [library]  (an experimental build of the library for A/B runs)."""
import sys, os, torch
sys.path.insert(0, os.getcwd())
from gtg_vae_amd import _lib
if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
from gtg_vae_amd import mst_graph_batched
def timed(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
g = torch.Generator().manual_seed(1)
for N in (10, 32, 48, 64):
    M = N * (N + 1) // 2
    rows = torch.rand(65536, M, generator=g).cuda()
    full = torch.rand(8192, N, N, generator=g).cuda()
    print("N=%d triu %.4f ms  full(8192) %.4f ms" % (N, timed(lambda: mst_graph_batched(rows, N)), timed(lambda: mst_graph_batched(full, N))))
