#!/usr/bin/env python3
"""Time the batch-sized GEMM shapes of the latent head and the decoder
(M = batch) with HIP events; the tile configuration is chosen by the library
(the library picks the tile configuration)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gtg_vae_amd import functional as Fn

torch.manual_seed(0)
dev = "cuda"
shapes = [  # (name, A shape, B shape, a_kc, b_kc)
    ("nt 512x512x512", (512, 512), (512, 512), True, True),
    ("nt 512x2080x512", (512, 512), (2080, 512), True, True),
    ("nn 512x512x512", (512, 512), (512, 512), True, False),
    ("nn 512x512x2080", (512, 2080), (2080, 512), True, False),
    ("tn 512x512x512", (512, 512), (512, 512), False, False),
    ("tn 2080x512x512", (512, 2080), (512, 512), False, False),
    ("tn 1024x64x20800", (20800, 1024), (20800, 64), False, False),
    ("nt 512x2048x512", (512, 512), (2048, 512), True, True),
    ("nt 512x2112x512", (512, 512), (2112, 512), True, True),
    ("nt 512x1024x512", (512, 512), (1024, 512), True, True),
    ("nt 512x4096x512", (512, 512), (4096, 512), True, True),
    ("tn 2048x512x512", (512, 2048), (512, 512), False, False),
    ("nn 512x512x2048", (512, 2048), (2048, 512), True, False),
]
for name, sa, sb, akc, bkc in shapes:
    A = torch.randn(sa, device=dev)
    B = torch.randn(sb, device=dev)
    ref = None
    for _ in range(3):
        out = Fn.gemm(A, B, akc, bkc)
    a64, b64 = A.double(), B.double()
    ref = (a64 @ b64.T) if (akc and bkc) else (a64 @ b64) if akc else (a64.T @ b64)
    err = (out.double() - ref).abs().max().item() / ref.abs().max().item()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 50
    e0.record()
    for _ in range(reps):
        Fn.gemm(A, B, akc, bkc, out=out)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    m, n = out.shape
    k = sa[1] if akc else sa[0]
    print("%-20s %7.1f us  %6.1f TF  relerr %.1e" % (name, us, 2.0 * m * n * k / us / 1e6, err), flush=True)
