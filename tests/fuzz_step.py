#!/usr/bin/env python3
"""Randomised parity sweep: the fused training step (losses, every gradient) against the CPU oracle over
random (B, N, d, hid), eager and HIP-graph replay.  Usage: fuzz_step.py [cases] [seed] [big|small] [matrix mode]
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
mode = sys.argv[4] if len(sys.argv) > 4 else None   # None = the library's default


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
    tr = GraphVAETrainer(N, d, hid, N, device="cuda", init=params, matrix_mode=mode)
    with torch.no_grad():
        _, nodes = O.encode(params, None, batch.edge_index, B, N, return_nodes=True)
    for use_graph in (False, True):
        res = tr.step(batch.to("cuda"), eps.cuda(), kl_w=kl_w, update=False, use_graph=use_graph)
        assert tr.edges_ok()
        # The max-pool's winners against the oracle's: two rows of a graph that tie to within rounding may be
        # picked differently; such a flip (legitimate only on a tie of 1e-6 relative, and only for a small
        # fraction of the arg-maxes) moves one graph's share of one column's gradient to another row, so the
        # gradients are then compared with the oracle routed through the library's winners.
        want = p
        won = tr.argmax[:B].long().cpu()
        top2 = nodes.topk(min(2, N), dim=1).values
        flipped = (nodes.gather(1, won.view(B, 1, d)).squeeze(1) != top2[:, 0]).nonzero()
        if len(flipped):
            # a lead is measured against the graph's largest node value: rounding acts on the products' size, and a
            # maximum that happens to lie near zero would make a lead relative to itself meaningless
            leads = [(top2[b_, 0, c_] - nodes[b_, won[b_, c_], c_]).abs().item() / max(nodes[b_].abs().max().item(), 1e-30)
                     for b_, c_ in flipped.tolist()]
            # (structurally equivalent nodes -- two leaves of one parent -- give exactly equal rows, which the
            # oracle's scatter sums may separate by an ulp: such twins account for most flips at small N)
            assert max(leads) <= 1e-6 and len(flipped) <= 8 + B * d // 1000, (
                "case %d B=%d N=%d d=%d hid=%d: %d flips, largest lead %.2e" % (c, B, N, d, hid, len(flipped), max(leads)))
            want = {k: v.clone().requires_grad_(True) for k, v in params.items()}
            O.graph_step_loss(want, None, batch.edge_index, batch.adj_padded, eps, kl_w, B, N, won)[0].backward()
            print("    (%d winner flip(s) on ties, largest lead %.2e: gradients compared through the library's winners)" % (
                len(flipped), max(leads)))
        e_rec = abs(res["recon"].item() - rec.item()) / abs(rec.item())
        e_kl = abs(res["kl"].item() - kl.item()) / (abs(kl.item()) + 1e-7)
        e_g = max(relerr(tr.named_grads()[k], want[k].grad) for k in want if want[k].grad.abs().max() > 0)
        worst = max(worst, e_rec, e_kl, e_g)
        status = "ok" if (e_rec < 1e-5 and e_kl < 1e-5 and e_g < 1e-4) else "FAIL"
        print("case %2d B=%3d N=%2d d=%3d hid=%3d graph=%d  recon %.1e kl %.1e grads %.1e %s" % (
            c, B, N, d, hid, use_graph, e_rec, e_kl, e_g, status), flush=True)
        if status == "FAIL":
            bad = {k: relerr(tr.named_grads()[k], want[k].grad) for k in want}
            print("   ", bad)
            sys.exit(1)
print("worst relative error %.2e over %d cases" % (worst, cases))
