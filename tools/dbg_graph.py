#!/usr/bin/env python3
"""Debug aid: eager step vs HIP-graph replay on the bench workload; prints the
adjacency error flag and the losses after every step."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gtg_vae_amd.data import synth_graph_batch
from gtg_vae_amd.trainer import GraphVAETrainer

N, B, d, hid = 64, int(os.environ.get("DBG_B", "512")), 512, 512
dev = torch.device("cuda", 0)
tr = GraphVAETrainer(N, d, hid, N, device=dev, seed=0, max_batch=B, max_edges=B * N * 4)
batches = [synth_graph_batch(B, N, seed=1234 + i).to(dev) for i in range(2)]
gen = torch.Generator().manual_seed(4321)
eps = [torch.randn(B, d, generator=gen).to(dev) for _ in range(2)]
tr.set_kl_weight(0.5)
census = os.environ.get("DBG_CENSUS", "1") == "1"
if census:
    for i in range(2):
        tr.load_batch(batches[i], eps[i])
        tr.run_loaded(update=False)
        torch.cuda.synchronize()
        print("census", i, "err", tr.err.item(), "rows", tr.meta[2 * B + 1].item(), tr.scalars.tolist(), flush=True)
for i in range(6):
    tr.load_batch(batches[i % 2], eps[i % 2])
    tr.run_loaded(update=True, use_graph=True)
    torch.cuda.synchronize()
    print("graph step", i, "err", tr.err.item(), "rows", tr.meta[2 * B + 1].item(), tr.scalars.tolist(), flush=True)

# after the last replay (batch index 1): compare the device adjacency counts with a host recount
b = batches[1]
ei = b.edge_index.cpu()
keep = ei[0] != ei[1]
src, dst = ei[0][keep], ei[1][keep]
exp = torch.zeros(B * N * N, dtype=torch.int32)
exp.index_put_(((dst // N) * N * N + (dst % N) * N + (src % N),), torch.ones_like(src, dtype=torch.int32), accumulate=True)
got = tr.counts.cpu()
print("counts equal:", bool((got == exp).all()), "sum got", int(got.sum()), "sum exp", int(exp.sum()))
nb = tr.meta[:B].cpu()
print("n_b min/max/mean", int(nb.min()), int(nb.max()), float(nb.float().mean()))
print("ei tail zero:", bool((tr.ei[:, ei.shape[1]:] == 0).all()), "ei head equal:", bool((tr.ei[:, :ei.shape[1]].cpu() == ei).all()))
bad = torch.nonzero(got != exp).flatten()
print("mismatches", bad.numel(), "first", bad[:5].tolist(), "last", bad[-5:].tolist())
print("got ints", got[bad[:8]].tolist())
print("as float", got[bad[:8]].view(torch.float32).tolist())
print("ws bytes", tr.ws.numel(), "ptrs counts %x ws %x adj %x h1 %x adjc %x" % (tr.counts.data_ptr(), tr.ws.data_ptr(), tr.adj.data_ptr(), tr.h1.data_ptr(), tr.adjc.data_ptr()))
