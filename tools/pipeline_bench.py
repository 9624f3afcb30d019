#!/usr/bin/env python3
"""Training throughput INCLUDING the host side (collate, PCIe copy) at the benchmark configuration:
inline copies vs the prefetching loader.  The headline bench keeps inputs resident in HBM; this is the
host-buffer-inclusive rate."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from gtg_vae_amd.data import DataLoader, DeviceDataset, DevicePrefetcher, synth_graph_list
from gtg_vae_amd.trainer import GraphVAETrainer

N, B, d, hid = 64, 512, 512, 512
data = synth_graph_list(B * 24, N, seed=1)
loader = DataLoader(data, batch_size=B, nmax=N)
tr = GraphVAETrainer(N, d, hid, N, device="cuda", seed=0, max_batch=B, max_edges=B * N * 4)
tr.set_kl_weight(0.5)
gen = torch.Generator().manual_seed(0)
eps = torch.randn(B, d, generator=gen).cuda()
for name, ld in (("inline copies", loader), ("prefetch depth 2", DevicePrefetcher(loader, "cuda", 2)),
                 ("prefetch depth 4", DevicePrefetcher(loader, "cuda", 4))):
    for epoch in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 0
        for batch in ld:
            tr.load_batch(batch if batch.edge_index.is_cuda else batch.to("cuda"), eps)
            tr.run_loaded(update=True, use_graph=True)
            n += batch.num_graphs
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print("%-18s %.0f graphs/s  (%.2f ms per %d-graph batch, host collate included)" % (name, n / dt, dt / len(ld) * 1e3, B), flush=True)

ds = DeviceDataset(data, N, "cuda")
for epoch in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    n = 0
    for ids in ds.batches(B):
        tr.load_indices(ds, ids, eps[: len(ids)])
        tr.run_loaded(update=True, use_graph=True)
        n += len(ids)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
print("%-18s %.0f graphs/s  (%.2f ms per %d-graph batch; dataset resident in HBM, batches gathered on the device)" % (
    "device dataset", n / dt, dt / (len(data) // B) * 1e3, B), flush=True)
assert tr.edges_ok()
