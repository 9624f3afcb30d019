"""Data-parallel path on CPU: two gloo ranks shard one batch by graph, reduce
their flat gradient buckets, and must reproduce the single-process gradient."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, B, N, d, hid, out_path):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    torch.set_num_threads(2)
    from gtg_vae_amd.data import synth_graph_batch
    from gtg_vae_amd.parallel import GradBuckets, init_process_group, shard_range, shard_weight
    from gtg_vae_amd.train import slice_batch
    from oracle import graphvae_oracle as O

    r, _, w = init_process_group("gloo")
    assert (r, w) == (rank, world)
    full = synth_graph_batch(B, N, seed=5)
    eps = torch.randn(B, d, generator=torch.Generator().manual_seed(1))
    lo, hi = shard_range(B, rank, world)
    mine = slice_batch(full, lo, hi)
    p = {k: v.clone().requires_grad_(True) for k, v in O.init_params(N, d, hid, N, seed=2).items()}
    names = sorted(p)
    if hi > lo:
        loss, _, _, _ = O.graph_step_loss(p, None, mine.edge_index, mine.adj_padded, eps[lo:hi], 0.5, hi - lo, N)
        loss.backward()
        # this shard's mean gradient, weighted by its share of the global batch (x world: Adam divides by it)
        flat = torch.cat([p[k].grad.flatten() for k in names]) * shard_weight(hi - lo, mine.global_graphs, world)
    else:  # no graph on this rank: zeros, but the same collectives
        flat = torch.zeros(sum(p[k].numel() for k in names))
    split = flat.numel() // 3
    buckets = GradBuckets(flat, [0, split, flat.numel()])
    assert buckets.world == world
    buckets.reduce(0)
    buckets.reduce(1)
    buckets.finish()
    flat *= buckets.scale
    if rank == 0:
        q = {k: v.clone().requires_grad_(True) for k, v in O.init_params(N, d, hid, N, seed=2).items()}
        l2, _, _, _ = O.graph_step_loss(q, None, full.edge_index, full.adj_padded, eps, 0.5, B, N)
        l2.backward()
        want = torch.cat([q[k].grad.flatten() for k in names])
        torch.save({"got": flat, "want": want}, out_path)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("B", [8, 5, 1])   # equal shards; 3 + 2 graphs; 1 + 0 graphs
def test_two_rank_bucketed_allreduce_equals_single_process_gradient(tmp_path, B):
    out = str(tmp_path / "res.pt")
    mp.spawn(_worker, args=(2, _free_port(), B, 10, 16, 32, out), nprocs=2, join=True)
    res = torch.load(out)
    err = (res["got"] - res["want"]).abs().max().item() / res["want"].abs().max().item()
    assert err < 1e-5, err
