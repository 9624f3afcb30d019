#!/usr/bin/env python3
"""A/B of two builds of the library on one LSTM direction (T=8, B=256, I=512, H=1024, every matrix mode): sha1 of
the outputs and gradients, and the time of forward + backward.   python tools/lstm_ab.py [library]"""
import hashlib
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gtg_vae_amd import _lib
from gtg_vae_amd import functional as Fn

if len(sys.argv) > 1:
    _lib.LIB_PATH = os.path.abspath(sys.argv[1])
T, B, I, H = 8, 256, 512, 1024
torch.manual_seed(0)
mod = torch.nn.LSTM(I, H, 1).cuda()
x = torch.randn(T, B, I, device="cuda", requires_grad=True)
gout = torch.randn(T, B, H, device="cuda")
P = [mod.weight_ih_l0, mod.weight_hh_l0, mod.bias_ih_l0, mod.bias_hh_l0]
for mode in ("f16x2", "bf16x3", "f32"):
    for rev in (False, True):
        for p in P + [x]:
            p.grad = None
        hs, hn, cn = Fn.lstm(x, *P, None, None, rev, mode)
        ((hs * gout).sum() + hn.sum()).backward()
        h = hashlib.sha1()
        for t in [hs, hn, cn, x.grad] + [p.grad for p in P]:
            h.update(t.detach().cpu().numpy().tobytes())
        print(mode, rev, h.hexdigest())
T = 40
x = torch.randn(T, B, I, device="cuda", requires_grad=True)
gout = torch.randn(T, B, H, device="cuda")


def fb():
    for p in P + [x]:
        p.grad = None
    (Fn.lstm(x, *P)[0] * gout).sum().backward()


for _ in range(3):
    fb()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    fb()
e1.record()
torch.cuda.synchronize()
print("forward + backward, T=40: %.3f ms" % (e0.elapsed_time(e1) / 10))
