#!/usr/bin/env python3
"""Accuracy of the selected matrix pipe (GTG_MATRIX_MODE) against float64 on the three layouts."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

MODE = os.environ.get("GTG_MATRIX_MODE", "f32")  # matrix pipe under test: f32 | bf16x3 | bf16
from gtg_vae_amd import functional as Fn
torch.manual_seed(0)
for name, sa, sb, akc, bkc in [("nt", (8192, 1024), (512, 1024), True, True), ("nn", (4096, 512), (512, 1024), True, False),
                               ("tn", (4096, 2048), (4096, 2048), False, False), ("nt ragged", (5000, 333), (700, 333), True, True),
                               ("nn ragged", (5001, 77), (77, 701), True, False), ("tn ragged", (1003, 2100), (1003, 1801), False, False)]:
    A = torch.randn(*sa, device="cuda"); B = torch.randn(*sb, device="cuda")
    got = Fn.gemm(A, B, akc, bkc, split_k=False, matrix_mode=MODE)
    a, b = A.double(), B.double()
    want = (a @ b.T) if (akc and bkc) else (a @ b) if akc else (a.T @ b)
    err = (got.double() - want).abs().max().item() / want.abs().max().item()
    print("%-10s max err / max |C| = %.2e" % (name, err), flush=True)
