"""Worker of test_data_parallel_step_on_gpu_equals_full_batch_oracle: launched twice by torch.distributed.run
(gloo backend, both ranks on the one GPU).  Each rank trains on its half of every batch through the HIP path
(eager, then HIP-graph replay); after each step rank 0 compares the parameters with the CPU oracle's
full-batch Adam step.  Global batches of 12 (equal shards), 13 (shards of 7 and 6: each rank's gradient is
weighted by B_local * world / B_global) and 1 (rank 1 has no graph at all and contributes zeros).  Prints one
JSON line; exit status 1 on a mismatch."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.distributed as dist

from gtg_vae_amd.data import synth_graph_batch
from gtg_vae_amd.parallel import init_process_group, shard_range
from gtg_vae_amd.train import slice_batch
from gtg_vae_amd.trainer import GraphVAETrainer
from oracle import graphvae_oracle as O

rank, _, world = init_process_group()
N, d, hid = 10, 16, 32
params = O.init_params(N, d, hid, N, seed=2)
tr = GraphVAETrainer(N, d, hid, N, device="cuda", init=params)
assert tr.world == world == 2
p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
m = {k: torch.zeros_like(v) for k, v in p.items()}
v2 = {k: torch.zeros_like(v) for k, v in p.items()}
worst = 0.0
for step, B in enumerate((12, 12, 12, 12, 13, 13, 13, 1, 1, 12), start=1):
    full = synth_graph_batch(B, N, seed=40 + step)
    eps = torch.randn(B, d, generator=torch.Generator().manual_seed(step))
    lo, hi = shard_range(B, rank, world)
    mine = slice_batch(full, lo, hi)
    assert mine.global_graphs == B and mine.num_graphs == hi - lo
    tr.step(mine.to("cuda"), eps[lo:hi].cuda(), kl_w=0.1 * step, update=True, use_graph=step not in (1, 2, 5))
    assert tr.edges_ok()
    for t in p.values():
        t.grad = None
    loss, _, _, _ = O.graph_step_loss(p, None, full.edge_index, full.adj_padded, eps, 0.1 * step, B, N)
    loss.backward()
    with torch.no_grad():
        for k in p:
            O.adam_step(p[k], p[k].grad, m[k], v2[k], step)
    for k in p:
        worst = max(worst, (tr.named_params()[k].cpu() - p[k].detach()).abs().max().item())
# every rank holds the same parameters
flat = tr.flat_p.detach().cpu().clone()
other = flat.clone()
dist.broadcast(other, src=0)
same = bool(torch.equal(flat, other))
dist.barrier()
if rank == 0:
    print(json.dumps({"worst_abs_param_error": worst, "ranks_identical": same}))
ok = worst < 2e-5 and same
dist.destroy_process_group()
sys.exit(0 if ok else 1)
