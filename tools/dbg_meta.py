import sys, os
sys.path.insert(0, '/root/repo')
import torch
from gtg_vae_amd.trainer import GraphVAETrainer
from gtg_vae_amd.data import synth_graph_batch
from gtg_vae_amd import _lib
B,N,d=512,64,512
tr = GraphVAETrainer(N, d, 512, N, device="cuda", seed=0, max_batch=B, max_edges=B*N*4)
b = synth_graph_batch(B, N, seed=1234).to("cuda")
eps = torch.randn(B, d).cuda()
tr.step(b, eps, 0.5)
torch.cuda.synchronize()
m = tr.meta.cpu()
print("n_b mean", m[:B].float().mean().item(), "R", m[2*B].item(), "off[-3:]", m[2*B-2:2*B+1].tolist(), "num_nodes sum", b.num_nodes.sum().item())
_lib.profile_enable(True)
for _ in range(5): tr.step(b, eps, 0.5)
torch.cuda.synchronize()
for r in _lib.profile_report(): print(r)
