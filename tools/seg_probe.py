#!/usr/bin/env python3
"""What the data-parallel form of the step costs on ONE GPU, before any byte crosses xGMI: the step as one HIP
graph, as the three graphs of the multi-rank path without collectives, and with the two real (one-rank RCCL)
asynchronous all-reduces between them.

    python tools/seg_probe.py
"""
import os, sys, time
sys.path.insert(0, os.getcwd())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from gtg_vae_amd.data import synth_graph_batch
from gtg_vae_amd.trainer import GraphVAETrainer
N, B, d, hid = 64, 512, 512, 512
batches = [synth_graph_batch(B, N, seed=1234 + i).to("cuda") for i in range(2)]
eps = [torch.randn(B, d).cuda() for _ in range(2)]
def run(tag, world, patch=None):
    tr = GraphVAETrainer(N, d, hid, N, device="cuda", seed=0, max_batch=B, max_edges=B * N * 4)
    tr.world = world; tr.buckets.world = world
    if patch: patch(tr)
    tr.set_kl_weight(0.5)
    for i in range(10):
        tr.load_batch(batches[i % 2], eps[i % 2]); tr.run_loaded(update=True, use_graph=True)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100):
        tr.load_batch(batches[i % 2], eps[i % 2]); tr.run_loaded(update=True, use_graph=True)
    torch.cuda.synchronize()
    print("%-40s %.3f ms/step" % (tag, (time.perf_counter() - t0) / 100 * 1e3), flush=True)
run("one graph", 1)
def noreduce(tr):
    tr.buckets.reduce = lambda i: None
    tr.buckets.finish = lambda: None
run("three graphs, no collectives", 2, noreduce)
run("three graphs + 2 all-reduces", 2)
dist.destroy_process_group()
