#!/usr/bin/env python3
"""Launch one GEMM shape a few times (for rocprofv3 --pmc passes).
usage: gemm_one.py {nt|nn|nnmask|tn|rows_nnmask|rows_nt|rows_planes} [reps]  |  gemm_one.py shape M N K {nt|nn|tn} [reps]
The rows_* shapes are the compact-row encoder GEMMs: 32832 allocated rows of which the
device-side count (20759, the mean of the benchmark's batches) are computed; rows_planes is
rows_nnmask with both operands arriving as three bf16 planes (the training step's form)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

MODE = os.environ.get("GTG_MATRIX_MODE", "f32")  # matrix pipe under test: f32 | bf16x3 | bf16

from gtg_vae_amd import functional as Fn

which = sys.argv[1] if len(sys.argv) > 1 else "nt"
if which == "shape":   # gemm_one.py shape M N K {nt|nn|tn} [reps]
    M_, N_, K_, lay = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
    reps = int(sys.argv[6]) if len(sys.argv) > 6 else 5
    torch.manual_seed(0)
    akc, bkc = lay != "tn", lay == "nt"
    A = torch.randn((M_, K_) if akc else (K_, M_), device="cuda")
    B = torch.randn((N_, K_) if bkc else (K_, N_), device="cuda")
    for _ in range(reps):
        out = Fn.gemm(A, B, akc, bkc, matrix_mode=MODE)
    torch.cuda.synchronize()
    print("done", which, out.shape)
    sys.exit(0)
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
BN, H, D = 32768, 1024, 512
torch.manual_seed(0)
if which == "nt":
    A, B, akc, bkc, epi, aux = torch.randn(BN, H, device="cuda"), torch.randn(D, H, device="cuda"), True, True, 0, None
elif which == "nn":
    A, B, akc, bkc, epi, aux = torch.randn(BN, D, device="cuda"), torch.randn(D, H, device="cuda"), True, False, 0, None
elif which == "nnmask":
    A, B, akc, bkc, epi = torch.randn(BN, D, device="cuda"), torch.randn(D, H, device="cuda"), True, False, 3
    aux = torch.randn(BN, H, device="cuda")
elif which.startswith("rows_"):
    pass
else:
    A, B, akc, bkc, epi, aux = torch.randn(BN, D, device="cuda"), torch.randn(BN, H, device="cuda"), False, False, 0, None
ROWS = 20759
if which.startswith("rows_"):
    M = 32832
    rd = torch.tensor([ROWS], dtype=torch.int32, device="cuda")
    if which == "rows_planes":
        A, W = torch.randn(M, D, device="cuda"), torch.randn(H, D, device="cuda")
        aux, out = torch.randn(M, H, device="cuda"), torch.empty(M, H, device="cuda")
        Ap, Wp = Fn.split_planes(A), Fn.split_planes(W)
        for _ in range(reps):
            Fn.gemm_planes_rows(Ap, Wp, rd, epilogue=3, aux=aux, out=out)
        torch.cuda.synchronize()
        print("done", which, out.shape)
        sys.exit(0)
    if which == "rows_nnmask":
        A, B, akc, bkc, epi = torch.randn(M, D, device="cuda"), torch.randn(D, H, device="cuda"), True, False, 3
        aux = torch.randn(M, H, device="cuda")
    else:
        A, B, akc, bkc, epi, aux = torch.randn(M, H, device="cuda"), torch.randn(D, H, device="cuda"), True, True, 0, None
    out = torch.empty((M, B.shape[0] if bkc else B.shape[1]), device="cuda")
    for _ in range(reps):
        Fn.gemm_rows(A, B, akc, bkc, rows_dev=rd, epilogue=epi, aux=aux, out=out, matrix_mode=MODE)
    torch.cuda.synchronize()
    print("done", which, out.shape)
    sys.exit(0)
for _ in range(reps):
    out = Fn.gemm(A, B, akc, bkc, epilogue=epi, aux=aux, matrix_mode=MODE)
torch.cuda.synchronize()
print("done", which, out.shape)
