#!/usr/bin/env python3
"""Times the text branch's gather / scatter / dropout kernels at the joint configuration (T = 40, B = 256, D = 512,
vocabulary 30 522, sentences of 8..40 tokens) against the stock PyTorch modules they replace.

    python tools/text_ops_bench.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gtg_vae_amd import functional as Fn
from gtg_vae_amd.data import collate, synth_graph_list

T, B, V, D = 40, 256, 30522, 512
gb = collate(synth_graph_list(B, 32, seed=1234, vocab=V, max_len=T), 32)
ids = gb.dec_seq.t().contiguous().cuda()
T = ids.shape[0]


def timed(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


out = {"rows": int(ids.numel()), "distinct_ids": int(ids.unique().numel())}
w = torch.nn.Parameter(torch.randn(V, D, device="cuda") * 0.1)
add = torch.randn(B, D, device="cuda", requires_grad=True)
dout = torch.randn(T, B, D, device="cuda")
for p in (0.0, 0.5):
    def hip_step():
        w.grad = None
        Fn.embed_dropout(ids, w, p, add).backward(dout)

    def hip_fwd():
        with torch.no_grad():
            Fn.embed_dropout(ids, w, p, add)

    emb = torch.nn.Embedding(V, D).cuda()
    drop = torch.nn.Dropout(p)

    def stock_step():
        emb.weight.grad = None
        (drop(emb(ids)) + add).backward(dout)

    def stock_fwd():
        with torch.no_grad():
            drop(emb(ids)) + add

    out["p=%.1f" % p] = {"hip_fwd_us": timed(hip_fwd), "hip_fwd_bwd_us": timed(hip_step),
                         "stock_fwd_us": timed(stock_fwd), "stock_fwd_bwd_us": timed(stock_step)}
x = torch.randn(T, B, 1024, device="cuda", requires_grad=True)
dy = torch.randn_like(x)
drop = torch.nn.Dropout(0.5)
out["dropout_1024"] = {"hip_fwd_bwd_us": timed(lambda: Fn.dropout(x, 0.5).backward(dy)),
                       "stock_fwd_bwd_us": timed(lambda: drop(x).backward(dy))}
from gtg_vae_amd.noise import word_drop
cpu_like = lambda: None
out["word_drop"] = {"hip_us": timed(lambda: word_drop(0, ids, 0.3))}
print(json.dumps(out))
