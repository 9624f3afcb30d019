#!/usr/bin/env python3
"""Projection + cross-entropy of the text decoder at the joint configuration's
size (T=40, B=256 -> 10 240 tokens, k=1024, vocab 30 522): the HIP path against
stock PyTorch (Linear + F.cross_entropy + argmax of softmax), forward+backward."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from gtg_vae_amd import functional as Fn
from gtg_vae_amd import _lib

T, B, K, V = 40, 256, 1024, 30522
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dev = torch.device("cuda")
g = torch.Generator(device="cpu").manual_seed(3)
lens = torch.randint(8, T + 1, (B,), generator=g)
tgt = torch.randint(1000, V, (T, B), generator=g)
tgt[torch.arange(T)[:, None] >= lens[None, :]] = 0
tgt = tgt.to(dev)
H = torch.randn(T * B, K, device=dev, requires_grad=True)
W = (torch.rand(V, K, device=dev) * 0.2 - 0.1).requires_grad_()
b = torch.zeros(V, device=dev, requires_grad=True)
kept = int((tgt != 0).sum())


def timed(fn):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def stock():
    H.grad = W.grad = b.grad = None
    logits = F.linear(H, W, b)
    loss = F.cross_entropy(logits, tgt.view(-1), ignore_index=0, reduction="none").view(T, B)
    pred = torch.argmax(torch.softmax(logits.view(T, B, V), 2), 2)
    loss.sum(dim=0).mean().backward()
    return loss, pred


def ours(mode):
    def run():
        H.grad = W.grad = b.grad = None
        loss, pred = Fn.proj_ce(H, W, b, tgt.view(-1), 0, mode)
        loss.view(T, B).sum(dim=0).mean().backward()
        return loss, pred
    return run


out = {"tokens": T * B, "kept_tokens": kept, "k": K, "vocab": V}
out["stock_torch_ms"] = timed(stock)
for mode in ("f32", "bf16x3", "f16x2", "bf16"):
    out[f"hip_{mode}_ms"] = timed(ours(mode))
flops = 3 * 2.0 * kept * K * V
out["algorithmic_tflop_per_step"] = flops / 1e12
for mode in ("f32", "bf16x3", "f16x2", "bf16"):
    out[f"hip_{mode}_tflops"] = flops / out[f"hip_{mode}_ms"] / 1e9
out["stock_tflops_on_all_rows"] = 3 * 2.0 * T * B * K * V / out["stock_torch_ms"] / 1e9
_lib.profile_enable(True)
ours("f16x2")()
torch.cuda.synchronize()
out["kernels_f16x2"] = [(k, a, b_, c, n, round(ms, 4)) for k, a, b_, c, n, ms in _lib.profile_report()]
_lib.profile_enable(False)
print(json.dumps(out))
