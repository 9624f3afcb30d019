#!/usr/bin/env python3
"""Time the first GCN layer on compact rows (sparse row kernel inside
gtg_encoder_compact_fwd) through the library's per-launch profile."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gtg_vae_amd import _lib
from gtg_vae_amd import functional as Fn
from gtg_vae_amd.data import synth_graph_batch

B, N, d = 512, 64, 512
gb = synth_graph_batch(B, N, seed=1234).to("cuda")
torch.manual_seed(0)
W1 = torch.randn(2 * d, N, device="cuda") * 0.1
b1 = torch.randn(2 * d, device="cuda") * 0.1
W2 = torch.randn(d, 2 * d, device="cuda") * 0.05
b2 = torch.zeros(d, device="cuda")
adj, extents = Fn.build_adj(gb.edge_index, B, N, return_extents=True)
for _ in range(3):
    Fn.EncoderFn.apply(adj, None, W1, b1, W2, b2, extents, "bf16x3")
torch.cuda.synchronize()
_lib.profile_enable(True)
for _ in range(20):
    Fn.EncoderFn.apply(adj, None, W1, b1, W2, b2, extents, "bf16x3")
torch.cuda.synchronize()
for k, a, b, c, n, ms in _lib.profile_report():
    print("%-16s %6d %6d %6d  x%d  %.1f us" % (k, a, b, c, n, ms / n * 1e3))
