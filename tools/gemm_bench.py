#!/usr/bin/env python3
"""Time the fp32 MFMA GEMM on the shapes of the config-C training step
(HIP-event timing inside the library).  Usage: python tools/gemm_bench.py [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

MODE = os.environ.get("GTG_MATRIX_MODE", "f32")  # matrix pipe under test: f32 | bf16x3 | bf16

from gtg_vae_amd import _lib
from gtg_vae_amd import functional as Fn

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
BN, H, D = 32768, 1024, 512
shapes = [
    ("NT T2=H1*W2^T", (BN, H), (D, H), True, True, 0, None),
    ("NN dZ1=dT2*W2 (mask)", (BN, D), (D, H), True, False, 3, (BN, H)),
    ("NN dZ1 plain", (BN, D), (D, H), True, False, 0, None),
    ("TN dW2=dT2^T*H1", (BN, D), (BN, H), False, False, 0, None),
    ("TN dW1=dZ1^T*A", (BN, H), (BN, 64), False, False, 0, None),
    ("NT dec2", (512, 512), (2080, 512), True, True, 2, None),
    ("TN dD2", (512, 2080), (512, 512), False, False, 0, None),
    ("NN dy1", (512, 2080), (2080, 512), True, False, 0, None),
]
torch.manual_seed(0)
for name, sa, sb, akc, bkc, epi, saux in shapes:
    A = torch.randn(*sa, device="cuda")
    B = torch.randn(*sb, device="cuda")
    aux = torch.randn(*saux, device="cuda") if saux else None
    M, K = (sa[0], sa[1]) if akc else (sa[1], sa[0])
    N = sb[0] if bkc else sb[1]
    out = torch.empty(M, N, device="cuda")
    bias = torch.zeros(N, device="cuda") if epi in (1, 2) else None
    for _ in range(3):
        Fn.gemm(A, B, akc, bkc, bias=bias, epilogue=epi, aux=aux, out=out, matrix_mode=MODE)
    torch.cuda.synchronize()
    t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        Fn.gemm(A, B, akc, bkc, bias=bias, epilogue=epi, aux=aux, out=out, matrix_mode=MODE)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / reps
    print("%-26s M=%6d N=%5d K=%6d  %8.3f ms  %7.1f TFLOP/s" % (name, M, N, K, ms, 2.0 * M * N * K / ms / 1e9))
