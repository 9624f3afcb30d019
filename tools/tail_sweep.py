#!/usr/bin/env python3
"""Time the row-ragged 128x128 GEMM as a function of the device-side row count
(tiles per round of resident blocks)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

MODE = os.environ.get("GTG_MATRIX_MODE", "f32")  # matrix pipe under test: f32 | bf16x3 | bf16

from gtg_vae_amd import functional as Fn

which = sys.argv[1] if len(sys.argv) > 1 else "nt"
M = 32832
if which == "nt":
    N, K = 512, 1024
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda")
    bkc = True
else:
    N, K = 1024, 512
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(K, N, device="cuda")
    bkc = False
out = torch.empty((M, N), device="cuda")
tiles_n = N // 128
print("cus", torch.cuda.get_device_properties(0).multi_processor_count)
for tiles in [256, 384, 512, 520, 576, 640, 652, 704, 768, 896, 1024, 1100, 1304, 1536]:
    rows = tiles // tiles_n * 128
    if rows > M:
        break
    rd = torch.tensor([rows], dtype=torch.int32, device="cuda")
    for _ in range(3):
        Fn.gemm_rows(A, B, True, bkc, rows_dev=rd, out=out, matrix_mode=MODE)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 20
    e0.record()
    for _ in range(reps):
        Fn.gemm_rows(A, B, True, bkc, rows_dev=rd, out=out, matrix_mode=MODE)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    print("%s tiles %5d rows %6d  %7.1f us  %6.1f TF" % (which, tiles // tiles_n * tiles_n, rows, us, 2.0 * rows * N * K / us / 1e6), flush=True)
