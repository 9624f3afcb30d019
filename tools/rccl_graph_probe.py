#!/usr/bin/env python3
"""One-rank RCCL rehearsal: the trainer's graph-replayed step with a live NCCL(=RCCL) process group, its
watchdog thread, and real asynchronous all-reduces of both gradient buckets between the graph segments.
A single GPU cannot host two RCCL ranks, so this checks everything except the inter-GPU transport."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29533")
import torch
import torch.distributed as dist

torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from gtg_vae_amd.data import synth_graph_batch
from gtg_vae_amd.trainer import GraphVAETrainer

N, B, d, hid = 64, 512, 512, 512
tr = GraphVAETrainer(N, d, hid, N, device="cuda", seed=0, max_batch=B, max_edges=B * N * 4)
tr.world = 2                 # force the collective path (sum over one rank = identity; Adam then halves the gradient)
tr.buckets.world = 2
batches = [synth_graph_batch(B, N, seed=1234 + i).to("cuda") for i in range(2)]
eps = [torch.randn(B, d).cuda() for _ in range(2)]
tr.set_kl_weight(0.5)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 1
for mode in (False, True) * reps:
    for i in range(10):
        tr.load_batch(batches[i % 2], eps[i % 2])
        tr.run_loaded(update=True, use_graph=mode)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(200):
        tr.load_batch(batches[i % 2], eps[i % 2])
        tr.run_loaded(update=True, use_graph=mode)
    torch.cuda.synchronize()
    print("use_graph=%s: %.3f ms/step, recon %.5f kl %.5f" % (mode, (time.perf_counter() - t0) / 200 * 1e3,
                                                             tr.scalars[0].item(), tr.scalars[1].item()), flush=True)
dist.barrier()
dist.destroy_process_group()
print("ok")
