"""Worker of test_joint_data_parallel_step_on_gpu_equals_full_batch_oracle: launched twice by
torch.distributed.run (gloo backend, both ranks on the one GPU).  Each rank runs JointTrainer.step on its
shard of every global batch (evaluation mode, supplied eps); the three gradient buckets -- the vocabulary
projection first -- are all-reduced from the autograd hooks and one flat Adam step follows.  Every rank checks
the reduced gradient (sum over ranks / world) against the CPU oracle's full-batch gradient, and the
parameters against Adam applied on the CPU to that reduced gradient.  Global batches of 6 (equal shards), 5
(3 + 2 graphs) and 1 (rank 1 holds nothing).  Prints one JSON line; exit status 1 on a mismatch."""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, os.path.join(HERE, "golden"))
import torch
import torch.distributed as dist

from gtg_vae_amd import GTVAE
from gtg_vae_amd.data import collate, synth_graph_list
from gtg_vae_amd.joint_trainer import JointTrainer
from gtg_vae_amd.parallel import init_process_group, shard_range
from gtg_vae_amd.train import slice_batch
from make_text_golden import fill_parameters
from oracle import graphvae_oracle as O
from oracle import textvae_oracle as TO

rank, _, world = init_process_group()
N, d, hid, vocab, T = 10, 16, 32, 50, 8
torch.manual_seed(4)
model = GTVAE(N, d, hid, N, vocab, 0, text=True, train_graph_recon=True)
fill_parameters(model.text_vae, scale=0.05)
sd = {k: v.clone() for k, v in model.state_dict().items()}
tr = JointTrainer(N, d, hid, N, vocab, 0, device="cuda", init=sd, train_graph_recon=True)
tr.deterministic()
assert tr.world == world == 2
p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
m = {k: torch.zeros_like(v) for k, v in p.items()}
v2 = {k: torch.zeros_like(v) for k, v in p.items()}
worst = 0.0
worst_g = 0.0
for step, B in enumerate((6, 6, 5, 1, 6), start=1):
    full = collate(synth_graph_list(B, N, seed=60 + step, vocab=vocab, max_len=T), N)
    eps = torch.randn(B, d, generator=torch.Generator().manual_seed(step))
    lo, hi = shard_range(B, rank, world)
    mine = slice_batch(full, lo, hi).to("cuda")
    tr.step(mine, mine.dec_seq, eps[lo:hi].cuda(), 0.2 * step, update=True)
    for t in p.values():
        t.grad = None
    TO.joint_loss(p, full.edge_index, full.adj_padded, full.dec_seq, eps, 0.2 * step, B, N, 0, True)["loss"].backward()
    with torch.no_grad():
        for k in p:
            used = tr.named_grads()[k].detach().cpu() / world      # what this step's Adam consumed
            worst_g = max(worst_g, (used - p[k].grad).abs().max().item() / (p[k].grad.abs().max().item() + 1e-30))
            O.adam_step(p[k], used, m[k], v2[k], step)
    for k in p:
        worst = max(worst, (tr.named_params()[k].cpu() - p[k].detach()).abs().max().item())
flat = tr.flat_p.detach().cpu().clone()
other = flat.clone()
dist.broadcast(other, src=0)
same = bool(torch.equal(flat, other))
dist.barrier()
if rank == 0:
    print(json.dumps({"worst_abs_param_error": worst, "worst_rel_grad_error": worst_g, "ranks_identical": same}))
ok = worst < 1e-6 and worst_g < 1e-4 and same
dist.destroy_process_group()
sys.exit(0 if ok else 1)
