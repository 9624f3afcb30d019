#!/usr/bin/env python3
"""Time the shallow (K = 64) first-layer GEMM relu(adjc * W1^T + b1) on the compact rows."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gtg_vae_amd import functional as Fn

M, rows, N, K = 32832, 20759, 1024, 64
A = torch.randn(M, K, device="cuda")
B = torch.randn(N, K, device="cuda")
bias = torch.randn(N, device="cuda")
out = torch.empty((M, N), device="cuda")
rd = torch.tensor([rows], dtype=torch.int32, device="cuda")
for _ in range(5):
    Fn.gemm_rows(A, B, True, True, rows_dev=rd, bias=bias, epilogue=1, out=out)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(50):
    Fn.gemm_rows(A, B, True, True, rows_dev=rd, bias=bias, epilogue=1, out=out)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print("conv1 %.1f us  write %.2f TB/s" % (us, rows * N * 4 / us / 1e6))

# the same rows with the row count known to the host (no device-side extent read, no padded launch)
A2 = A[:rows].contiguous()
out2 = torch.empty((rows, N), device="cuda")
for _ in range(5):
    Fn.gemm(A2, B, True, True, bias=bias, epilogue=1, out=out2)
e0.record()
for _ in range(50):
    Fn.gemm(A2, B, True, True, bias=bias, epilogue=1, out=out2)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print("conv1 static rows %.1f us  write %.2f TB/s" % (us, rows * N * 4 / us / 1e6))
# no epilogue work: plain store
for _ in range(5):
    Fn.gemm(A2, B, True, True, out=out2)
e0.record()
for _ in range(50):
    Fn.gemm(A2, B, True, True, out=out2)
e1.record()
torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 50 * 1e3
print("conv1 static rows, plain store %.1f us" % us)
# K = 32 (one k step) and K = 128
for K2 in (32, 128, 256):
    A3 = torch.randn(rows, K2, device="cuda")
    B3 = torch.randn(N, K2, device="cuda")
    for _ in range(5):
        Fn.gemm(A3, B3, True, True, out=out2)
    e0.record()
    for _ in range(50):
        Fn.gemm(A3, B3, True, True, out=out2)
    e1.record()
    torch.cuda.synchronize()
    print("K=%d plain store %.1f us" % (K2, e0.elapsed_time(e1) / 50 * 1e3))
