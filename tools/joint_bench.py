#!/usr/bin/env python3
"""One training step of the joint graph+text model (gtvae.py:95-126 with the
text branch of text_vae.py) at batch 256: graphs as in the single-GPU graph
configuration (N=32, d=256), sentences of 8..40 tokens over the 30 522-word
vocabulary.  Times the step with the HIP projection + cross-entropy
(DAE.decode_loss) and with that one piece replaced by stock PyTorch
(Linear + cross_entropy + argmax of softmax), everything else equal."""
import json
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from gtg_vae_amd import GTVAE
from gtg_vae_amd.data import synth_graph_batch

B, N, d, hid, V, T = 256, 32, 256, 512, 30522, 40
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
dev = torch.device("cuda")
torch.manual_seed(0)
model = GTVAE(N, d, hid, N, V, 0, text=True).to(dev)
model.train()
opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.5, 0.999))
gb = synth_graph_batch(B, N, seed=1234).to(dev)
g = torch.Generator().manual_seed(7)
lens = torch.randint(8, T + 1, (B,), generator=g)
seq = torch.randint(1000, V, (B, T), generator=g)
seq[:, 0] = 101
seq[torch.arange(B), lens - 1] = 102
seq[torch.arange(T)[None, :] >= lens[:, None]] = 0
seq = seq.to(dev)
x = gb.dense_x()


def stock_decode_loss(self, z, input, targets, hidden=None, matrix_mode=None):
    logits, hidden = self.decode(z, input, hidden)
    loss = F.cross_entropy(logits.view(-1, logits.size(-1)), targets.flatten(), ignore_index=self.padidx,
                           reduction="none").view(targets.size())
    pred = torch.argmax(torch.softmax(logits, 2), 2)
    return loss, pred, hidden


def step():
    opt.zero_grad(set_to_none=True)
    res = model(x, gb.edge_index, gb.batch, gb.adj_padded, gb.gold_edges, seq, 0.3)
    res[0].backward()
    opt.step()
    return res


def timed():
    for _ in range(3):
        step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        res = step()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / steps, float(res[0])


out = {"workload": "joint graph+text step: batch 256, N=32, d=256, T<=40, vocab 30522, dropout and word noise on",
       "steps": steps, "kept_tokens": int((seq[:, 1:] != 0).sum())}
for mode in ("bf16x3", "f32", "bf16"):
    model.graph_vae.encoder.matrix_mode = model.text_vae.matrix_mode = mode
    ms, loss = timed()
    out[f"hip_{mode}"] = {"ms_per_step": ms, "graphs_per_s": B / ms * 1e3, "loss": loss}
model.text_vae.decode_loss = types.MethodType(stock_decode_loss, model.text_vae)
model.graph_vae.encoder.matrix_mode = model.text_vae.matrix_mode = "bf16x3"
ms, loss = timed()
out["stock_projection"] = {"ms_per_step": ms, "graphs_per_s": B / ms * 1e3, "loss": loss}
print(json.dumps(out))
