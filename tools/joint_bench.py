#!/usr/bin/env python3
"""Training step of the joint graph+text model (gtvae.py:95-126 with the text branch of text_vae.py,
BASELINE configs[4]) at batch 256 per GPU: graphs as in the single-GPU graph configuration (N=32, d=256),
synthetic sentences of 8..40 tokens over the 30 522-word vocabulary, dropout and word noise on.

Runs through JointTrainer (flat parameter / gradient buffers, one gtg_adam_step over every parameter,
bucketed all-reduce when launched with torchrun), and for comparison (a) the same module stepped by
torch.optim.Adam, (b) the same with the vocabulary projection + cross-entropy on stock PyTorch.

    python tools/joint_bench.py [steps]
    python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/joint_bench.py [steps]
"""
import json
import os
import sys
import types

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from gtg_vae_amd.data import collate, synth_graph_list
from gtg_vae_amd.joint_trainer import JointTrainer
from gtg_vae_amd.parallel import all_reduce_small, init_process_group, local_device_index

B, N, d, hid, V, T = 256, 32, 256, 512, 30522, 40
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
only_trainer = len(sys.argv) > 2 and sys.argv[2] == "trainer"   # just the default-mode trainer (for profiling)
if len(sys.argv) > 3:          # an experimental build of the library (A/B runs on one box)
    from gtg_vae_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[3])
rank, local_rank, world = init_process_group()
dev = torch.device("cuda", local_device_index(local_rank))
torch.cuda.set_device(dev)
gb = collate(synth_graph_list(B, N, seed=1234 + rank, vocab=V, max_len=T), N).to(dev)
eps = torch.randn(B, d, generator=torch.Generator().manual_seed(rank)).to(dev)


def timed(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        torch.distributed.barrier()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(steps):
        res = fn()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / steps
    if world > 1:
        ms = float(all_reduce_small(torch.tensor([ms], device=dev), torch.distributed.ReduceOp.MAX)[0])
    return ms, float(res)


out = {"workload": "joint graph+text step: batch 256 per GPU, N=32, d=256, T<=40, vocab 30522, dropout and word noise on",
       "steps": steps, "n_gpus": world, "kept_tokens": int((gb.dec_seq[:, 1:] != 0).sum())}
for mode in (("f16x2",) if only_trainer else ("f16x2", "bf16x3", "f32", "bf16")):
    tr = JointTrainer(N, d, hid, N, V, 0, device=dev, seed=0, matrix_mode=mode)
    tr.model.train()
    ms, loss = timed(lambda: tr.step(gb, gb.dec_seq, eps, 0.3, update=True)["loss"])
    out[f"trainer_{mode}"] = {"ms_per_step": ms, "graphs_per_s": world * B / ms * 1e3, "loss": loss,
                              "parameters": tr.numel}
    if mode != "bf16":
        del tr
        torch.cuda.empty_cache()

if world == 1 and not only_trainer:
    # the same module under torch.optim.Adam (per-tensor foreach kernels) instead of the flat fused step
    model = tr.model
    for p in model.parameters():
        p.grad = None
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=1e-3, betas=(0.5, 0.999))

    def torch_adam_step():
        opt.zero_grad(set_to_none=True)
        res = model(None, gb.edge_index, B, gb.adj_padded, gb.gold_bits, gb.dec_seq, 0.3, eps=eps)
        res[0].backward()
        opt.step()
        return res[0]

    model.graph_vae.encoder.matrix_mode = model.text_vae.matrix_mode = "f16x2"
    ms, loss = timed(torch_adam_step)
    out["torch_adam_f16x2"] = {"ms_per_step": ms, "graphs_per_s": B / ms * 1e3, "loss": loss}

    def stock_decode_loss(self, z, input, targets, hidden=None, matrix_mode=None):
        logits, hidden = self.decode(z, input, hidden)
        loss = F.cross_entropy(logits.view(-1, logits.size(-1)), targets.flatten(), ignore_index=self.padidx,
                               reduction="none").view(targets.size())
        pred = torch.argmax(torch.softmax(logits, 2), 2)
        return loss, pred, hidden

    model.text_vae.decode_loss = types.MethodType(stock_decode_loss, model.text_vae)
    ms, loss = timed(torch_adam_step)
    out["torch_adam_stock_projection"] = {"ms_per_step": ms, "graphs_per_s": B / ms * 1e3, "loss": loss}
if rank == 0:
    print(json.dumps(out))
if world > 1:
    torch.distributed.destroy_process_group()
