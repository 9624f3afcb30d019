#!/usr/bin/env python3
"""Randomised parity sweep: the fused training step (losses, every gradient) against the CPU oracle over
random (B, N, d, hid), eager and HIP-graph replay.  Usage: fuzz_step.py [cases] [seed] [big]
("big": batches and widths large enough for the sparse second-layer weight gradient and the paired GEMMs)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from gtg_vae_amd.data import synth_graph_batch
from gtg_vae_amd.trainer import GraphVAETrainer
from oracle import graphvae_oracle as O

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
big = len(sys.argv) > 3 and sys.argv[3] == "big"


def relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


worst = 0.0
for c in range(cases):
    N = int(rng.choice([1, 2, 3, 5, 8, 10, 17, 31, 32, 33, 48, 63, 64]))
    if big:
        B = int(rng.choice([257, 400, 512, 700, 1000]))
        d = int(rng.choice([96, 128, 160, 256, 288, 512]))
    else:
        B = int(rng.choice([1, 2, 3, 7, 16, 33, 64, 130, 257]))
        d = int(rng.choice([1, 2, 3, 4, 8, 12, 30, 32, 64, 100, 128]))
    hid = int(rng.choice([1, 5, 16, 64, 96, 130]))
    params = O.init_params(N, d, hid, N, seed=c)
    for k in params:                      # random biases too (the reference's initialisers leave some at zero)
        if k.endswith("bias"):
            params[k] = 0.1 * torch.randn(params[k].shape, generator=torch.Generator().manual_seed(c))
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    batch = synth_graph_batch(B, N, seed=100 + c)
    eps = torch.randn(B, d, generator=torch.Generator().manual_seed(c))
    kl_w = float(rng.uniform(0, 1))
    loss, rec, kl, _ = O.graph_step_loss(p, None, batch.edge_index, batch.adj_padded, eps, kl_w, B, N)
    loss.backward()
    tr = GraphVAETrainer(N, d, hid, N, device="cuda", init=params)
    for use_graph in (False, True):
        res = tr.step(batch.to("cuda"), eps.cuda(), kl_w=kl_w, update=False, use_graph=use_graph)
        assert tr.edges_ok()
        e_rec = abs(res["recon"].item() - rec.item()) / abs(rec.item())
        e_kl = abs(res["kl"].item() - kl.item()) / (abs(kl.item()) + 1e-7)
        e_g = max(relerr(tr.named_grads()[k], p[k].grad) for k in p if p[k].grad.abs().max() > 0)
        worst = max(worst, e_rec, e_kl, e_g)
        status = "ok" if (e_rec < 1e-5 and e_kl < 1e-5 and e_g < 1e-4) else "FAIL"
        print("case %2d B=%3d N=%2d d=%3d hid=%3d graph=%d  recon %.1e kl %.1e grads %.1e %s" % (
            c, B, N, d, hid, use_graph, e_rec, e_kl, e_g, status), flush=True)
        if status == "FAIL":
            bad = {k: relerr(tr.named_grads()[k], p[k].grad) for k in p}
            print("   ", bad)
            sys.exit(1)
print("worst relative error %.2e over %d cases" % (worst, cases))
