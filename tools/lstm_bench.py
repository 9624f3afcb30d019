#!/usr/bin/env python3
"""Time one LSTM direction at the text branch's size (text_vae.py:37-40: nn.LSTM(512, 1024), batch 256, 40
steps): the library's recurrence (functional.lstm) against the stock nn.LSTM module, forward and
forward + backward.    python tools/lstm_bench.py [T] [B] [library]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gtg_vae_amd import _lib
from gtg_vae_amd import functional as Fn

if len(sys.argv) > 3:          # an experimental build of the library (ablation runs)
    _lib.LIB_PATH = os.path.abspath(sys.argv[3])
    print("library:", _lib.LIB_PATH)

T = int(sys.argv[1]) if len(sys.argv) > 1 else 40
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
I, H = 512, 1024
torch.manual_seed(0)
mod = torch.nn.LSTM(I, H, 1).cuda()
x = torch.randn(T, B, I, device="cuda", requires_grad=True)
gout = torch.randn(T, B, H, device="cuda")
P = [mod.weight_ih_l0, mod.weight_hh_l0, mod.bias_ih_l0, mod.bias_hh_l0]


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


def hip_fwd():
    with torch.no_grad():
        return Fn.lstm(x, *P)[0]


def hip_fb():
    for p in P + [x]:
        p.grad = None
    (Fn.lstm(x, *P)[0] * gout).sum().backward()


def stock_fwd():
    with torch.no_grad():
        return mod(x)[0]


def stock_fb():
    for p in P + [x]:
        p.grad = None
    (mod(x)[0] * gout).sum().backward()


print("T=%d B=%d I=%d H=%d" % (T, B, I, H))
print("library  forward %.3f ms   forward+backward %.3f ms" % (timed(hip_fwd), timed(hip_fb)))
print("nn.LSTM  forward %.3f ms   forward+backward %.3f ms" % (timed(stock_fwd), timed(stock_fb)))
