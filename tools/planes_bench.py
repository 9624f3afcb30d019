#!/usr/bin/env python3
"""A/B of the two three-plane bf16 GEMMs on the encoder's backward shape (dZ1c = (dT2c W2) masked by relu:
20 759 live rows of 32 832, N = 1024, K = 512) and its forward shape (H1c W2^T: N = 512, K = 1024): operands
split in the GEMM's loop (GTG_MATRIX_BF16X3 on fp32 operands) against operands that arrive split."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gtg_vae_amd import _lib
from gtg_vae_amd import functional as Fn

if len(sys.argv) > 1:          # an experimental build of the library (A/B and ablation runs)
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
    print("library:", _lib.LIB_PATH)


def timed(fn, reps=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


for (M, N, K, rows, epi) in ((32832, 1024, 512, 20759, 3), (32832, 1024, 512, 20759, 0), (32832, 512, 1024, 20759, 0)):
    torch.manual_seed(0)
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda")
    aux = torch.randn(M, N, device="cuda") if epi == 3 else None
    rd = torch.tensor([rows], dtype=torch.int32, device="cuda")
    Ap, Wp = Fn.split_planes(A), Fn.split_planes(W)
    out = torch.zeros(M, N, device="cuda")
    t_loop = timed(lambda: Fn.gemm_rows(A, W, True, True, rd, epilogue=epi, aux=aux, out=out, matrix_mode="bf16x3"))
    t_pl = timed(lambda: Fn.gemm_planes_rows(Ap, Wp, rd, epilogue=epi, aux=aux, out=out))
    t_split = timed(lambda: Fn.split_planes(W))
    fl = 2.0 * rows * N * K
    print("M=%d(live %d) N=%d K=%d epi=%d: in-loop split %.4f ms (%.0f TF fp32-eq) | pre-split planes %.4f ms (%.0f TF, "
          "%.0f TF of bf16 products = %.3f of 2.5 PF) | weight split kernel %.4f ms" % (
              M, rows, N, K, epi, t_loop, fl / t_loop / 1e9, t_pl, fl / t_pl / 1e9, 6 * fl / t_pl / 1e9,
              6 * fl / t_pl / 1e9 / 2500, t_split))
