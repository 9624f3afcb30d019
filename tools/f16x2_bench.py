#!/usr/bin/env python3
"""The 128x128-tile GEMM on fp32 operands in the three split / rounded matrix modes, on the encoder's forward shape
and the text projection's three shapes: ms per call (the two-plane fp16 mode's figure includes its pass over the
operands for their largest magnitudes)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gtg_vae_amd import functional as Fn


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


SHAPES = [("NT", 20759, 512, 1024), ("NT", 6098, 30522, 1024), ("NN", 6098, 1024, 30522), ("TN", 30522, 1024, 6098),
          ("NT", 10240, 4096, 512), ("TN", 4096, 1024, 10240)]
for layout, M, N, K in SHAPES:
    a_kc, b_kc = layout != "TN", layout == "NT"
    A = torch.randn(M, K, device="cuda") if a_kc else torch.randn(K, M, device="cuda")
    B = torch.randn(N, K, device="cuda") if b_kc else torch.randn(K, N, device="cuda")
    out = torch.empty(M, N, device="cuda")
    res = {}
    for mode in ("bf16x3", "f16x2", "bf16", "f32"):
        res[mode] = timed(lambda: Fn.gemm(A, B, a_kc, b_kc, out=out, matrix_mode=mode))
    fl = 2.0 * M * N * K
    print("%s M=%d N=%d K=%d: " % (layout, M, N, K) + "  ".join("%s %.3f ms (%.0f TF)" % (m, t, fl / t / 1e9) for m, t in res.items()), flush=True)
