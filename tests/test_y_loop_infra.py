"""GPU tests of the plumbing AROUND the hot path: the train()/dev() loops, the prefetching loader and
its worker thread, the HBM-resident dataset, HIP-graph capture beside a loader thread, and the two-rank
data-parallel rehearsals (subprocesses).  Every comparison of a kernel with the oracle or a golden file
lives in the test_gpu_* files and test_text_branch.py, which pytest collects before this one: a failure
of the infrastructure here cannot keep a parity test from running under ``pytest -x``."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def test_train_and_dev_loop_runs_and_learns(capsys):
    """train.py:284-355 / 228-282 mirrored: per-epoch KL weight min(epoch/500, 1), batch-size-weighted epoch
    means, dev pass with the batched spanning-tree read-out; the printed field names are the reference's."""
    from gtg_vae_amd.data import DataLoader, synth_graph_list
    from gtg_vae_amd.train import dev, train
    from gtg_vae_amd.trainer import GraphVAETrainer

    N, d, hid = 10, 32, 64
    data = synth_graph_list(300, N, seed=1)
    trn, dv = DataLoader(data[:256], batch_size=64, nmax=N), DataLoader(data[256:], batch_size=64, nmax=N)
    tr = GraphVAETrainer(N, d, hid, N, device="cuda", seed=0, lr=1e-2)
    gen = torch.Generator().manual_seed(0)
    hist = train(trn, dv, tr, epochs=6, generator=gen)
    printed = capsys.readouterr().out
    for field in ("Epoch loss:", "Epoch graph_recon loss:", "Epoch joint_kl_loss:", "kl_anneal_w:", "DEV",
                  "Epoch edge_recall:", "Epoch edge_precision:"):
        assert field in printed
    assert hist[0]["kl"] >= 0 and all(np.isfinite(h["loss"]) for h in hist)
    assert hist[-1]["graph_recon"] < 0.8 * hist[0]["graph_recon"]       # it learns
    last = hist[-1]["dev"]
    assert 0.0 <= last["edge_recall"] <= 1.0 and 0.0 <= last["edge_precision"] <= 1.0
    # ragged tail: 300 - 256 = 44 dev graphs in one short batch
    again = dev(dv, tr, 0, 0.0, gen, quiet=True)
    assert np.isfinite(again["loss"])


def test_device_prefetcher_yields_the_loaders_batches_in_order():
    """The prefetching loader (worker thread, pinned memory, side-stream copies) hands over exactly the
    loader's batches, ragged tail included, and a training epoch through it equals the inline epoch."""
    from gtg_vae_amd.data import DataLoader, DevicePrefetcher, synth_graph_list
    from gtg_vae_amd.train import train
    from gtg_vae_amd.trainer import GraphVAETrainer

    N = 12
    data = synth_graph_list(150, N, seed=3)
    loader = DataLoader(data, batch_size=32, nmax=N)
    got = list(DevicePrefetcher(loader, "cuda", depth=3))
    want = list(loader)
    assert len(got) == len(want) == 5 and got[-1].num_graphs == 150 - 4 * 32
    for g, w in zip(got, want):
        assert g.edge_index.is_cuda and torch.equal(g.edge_index.cpu(), w.edge_index)
        assert torch.equal(g.adj_padded.cpu(), w.adj_padded) and torch.equal(g.gold_bits.cpu(), w.gold_bits)
        assert g.gold_edges == w.gold_edges and g.num_graphs == w.num_graphs
    runs = []
    # the third arm captures the step's HIP graphs while the prefetcher's worker thread pins and copies batches
    for wrap, use_graph in ((False, False), (True, False), (True, True)):
        tr = GraphVAETrainer(N, 16, 32, N, device="cuda", seed=1)
        ld = DevicePrefetcher(loader, "cuda") if wrap else loader
        hist = train(ld, None, tr, epochs=2, generator=torch.Generator().manual_seed(0), quiet=True,
                     use_graph=use_graph)
        runs.append((tr.flat_p.clone(), hist[-1]["loss"]))
    assert torch.equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]
    assert torch.equal(runs[0][0], runs[2][0]) and runs[0][1] == runs[2][1]

    # an exception in the worker surfaces in the consumer
    class Boom:
        dataset = []

        def __len__(self):
            return 1

        def __iter__(self):
            yield want[0]
            raise RuntimeError("loader failed")

    it = iter(DevicePrefetcher(Boom(), "cuda"))
    next(it)
    with pytest.raises(RuntimeError, match="loader failed"):
        next(it)


def test_device_dataset_gather_equals_host_collate():
    """Batches gathered on the device from the HBM-resident dataset equal the host collate (edge list up to the
    padding, targets) for sequential and permuted ids, ragged tail included; bad ids and overflow are flagged;
    training through load_indices equals training through load_batch bit for bit."""
    from gtg_vae_amd.data import DataLoader, DeviceDataset, collate, synth_graph_list
    from gtg_vae_amd.trainer import GraphVAETrainer

    N, B = 12, 32
    data = synth_graph_list(150, N, seed=3)
    ds = DeviceDataset(data, N, "cuda")
    tr = GraphVAETrainer(N, 16, 32, N, device="cuda", seed=1)
    gen = torch.Generator().manual_seed(0)
    for shuffle in (False, True):
        for ids in ds.batches(B, shuffle=shuffle, generator=gen):
            host = collate([data[i] for i in ids.tolist()], N)
            eps = torch.zeros(len(ids), 16, device="cuda")
            tr.load_indices(ds, ids, eps)
            torch.cuda.synchronize()
            assert tr.edges_ok()
            E = host.edge_index.shape[1]
            assert torch.equal(tr.ei[:, :E].cpu(), host.edge_index)
            assert bool((tr.ei[:, E:] == 0).all())
            assert torch.equal(tr.target[: len(ids)].cpu(), host.adj_padded)
    # flags
    tr.load_indices(ds, torch.tensor([0, 151, 2], device="cuda"), torch.zeros(3, 16, device="cuda"))
    assert not tr.edges_ok()
    tr.load_indices(ds, torch.tensor([0, 1, 2], device="cuda"), torch.zeros(3, 16, device="cuda"))
    assert tr.edges_ok()
    # same training trajectory as the host path
    runs = []
    for mode in ("host", "device"):
        t2 = GraphVAETrainer(N, 16, 32, N, device="cuda", seed=1, max_batch=B, max_edges=B * ds.max_edges_per_graph)
        g2 = torch.Generator().manual_seed(7)
        for epoch in range(2):
            for i, ids in enumerate(ds.batches(B)):
                eps = torch.randn(len(ids), 16, generator=g2).cuda()
                if mode == "host":
                    t2.load_batch(collate([data[j] for j in ids.tolist()], N).to("cuda"), eps)
                else:
                    t2.load_indices(ds, ids, eps)
                t2.run_loaded(update=True, use_graph=(mode == "device"))
        torch.cuda.synchronize()
        runs.append(t2.flat_p.clone())
    assert torch.equal(runs[0], runs[1])


def test_train_loop_over_the_device_resident_dataset():
    """train() fed by DeviceLoader (ids of an HBM-resident dataset) follows the host-loader trajectory exactly."""
    from gtg_vae_amd.data import DataLoader, DeviceDataset, DeviceLoader, synth_graph_list
    from gtg_vae_amd.train import train
    from gtg_vae_amd.trainer import GraphVAETrainer

    N = 10
    data = synth_graph_list(200, N, seed=5)
    runs = []
    for kind in ("host", "device"):
        tr = GraphVAETrainer(N, 16, 32, N, device="cuda", seed=2)
        ld = DataLoader(data, 64, N) if kind == "host" else DeviceLoader(DeviceDataset(data, N, "cuda"), 64)
        hist = train(ld, None, tr, epochs=3, generator=torch.Generator().manual_seed(1), quiet=True,
                     use_graph=(kind == "device"))
        assert tr.edges_ok()
        runs.append((tr.flat_p.clone(), [h["loss"] for h in hist]))
    assert torch.equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]


def test_data_parallel_step_on_gpu_equals_full_batch_oracle():
    """Two ranks (gloo, both on this GPU) each train on half of every batch through the HIP kernels, eager and
    graph replay, with the bucketed gradient all-reduce between the step's pieces: the parameters must follow
    the oracle's full-batch Adam steps and stay identical on both ranks (tests/dp_gpu_worker.py)."""
    import json
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, GTG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(here, "dp_gpu_worker.py")], env=env, capture_output=True, text=True, timeout=170)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["ranks_identical"] and out["worst_abs_param_error"] < 2e-5


def test_joint_train_and_dev_loop_runs(capsys):
    """train.py:284-355 / 228-282 for the joint model: the reference's printed fields, dropout and word
    noise on in training, spanning-tree statistics in the dev pass, and it learns the sentences."""
    import numpy as np

    from gtg_vae_amd.data import DataLoader, synth_graph_list
    from gtg_vae_amd.joint_trainer import JointTrainer
    from gtg_vae_amd.train import train_joint

    N, d, hid, vocab = 10, 32, 64, 40
    data = synth_graph_list(160, N, seed=1, vocab=vocab, max_len=10)
    trn, dv = DataLoader(data[:128], batch_size=32, nmax=N), DataLoader(data[128:], batch_size=32, nmax=N)
    tr = JointTrainer(N, d, hid, N, vocab, 0, device="cuda", seed=0, lr=3e-3, train_graph_recon=True)
    torch.manual_seed(0)
    hist = train_joint(trn, dv, tr, epochs=5, generator=torch.Generator().manual_seed(0))
    printed = capsys.readouterr().out
    for field in ("Epoch loss:", "Epoch graph_recon loss:", "Epoch text_recon_loss:", "Epoch joint_kl_loss:",
                  "Epoch text_acc:", "kl_anneal_w:", "DEV", "Epoch edge_recall:", "Epoch edge_precision:"):
        assert field in printed
    assert all(np.isfinite(h["loss"]) for h in hist)
    assert hist[-1]["text_recon"] < hist[0]["text_recon"] and hist[-1]["graph_recon"] < hist[0]["graph_recon"]
    last = hist[-1]["dev"]
    assert 0.0 <= last["edge_recall"] <= 1.0 and 0.0 <= last["text_acc"] <= 1.0


def test_joint_data_parallel_step_on_gpu_equals_full_batch_oracle():
    """Two ranks (gloo, both on this GPU) train the joint model on their shards of every batch, unequal and
    empty shards included; parameters must follow the oracle's full-batch Adam steps and stay identical on
    both ranks (tests/dp_joint_gpu_worker.py)."""
    import json
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    here = os.path.dirname(os.path.abspath(__file__))
    env = dict(os.environ, GTG_DIST_BACKEND="gloo", HSA_ENABLE_IPC_MODE_LEGACY="0")
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                          "--master-addr", "127.0.0.1", "--master-port", str(port),
                          os.path.join(here, "dp_joint_gpu_worker.py")], env=env, capture_output=True, text=True,
                         timeout=170)
    assert res.returncode == 0, res.stdout[-2000:] + res.stderr[-2000:]
    line = [l for l in res.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["ranks_identical"] and out["worst_abs_param_error"] < 1e-6 and out["worst_rel_grad_error"] < 1e-4


def _pool_handles(k=64):
    """Raw handles of the next ``k`` streams PyTorch hands out (they come round-robin from a fixed pool)."""
    return [int(torch.cuda.Stream().cuda_stream) for _ in range(k)]


def test_owned_streams_are_outside_the_stream_pool():
    """The capture, copy and LSTM side streams are created by the library, once per device: pairwise distinct,
    none of them the default stream, none of them a handle that torch.cuda.Stream() (and therefore a
    collective's stream, drawn from the same pool) can ever return; asking twice gives the same object."""
    from gtg_vae_amd import _lib
    from gtg_vae_amd.streams import ROLES, owned_handles, owned_stream

    streams = {r: owned_stream("cuda", r) for r in ROLES}
    for r in ROLES:
        assert owned_stream("cuda", r) is streams[r]
    handles = owned_handles("cuda")
    assert set(handles) == set(ROLES) and len(set(handles.values())) == len(ROLES)
    assert 0 not in handles.values()
    assert int(torch.cuda.default_stream().cuda_stream) not in handles.values()
    pool = _pool_handles(96)
    assert len(set(pool)) < len(pool)                 # the pool does recycle: that is the hazard
    assert not set(pool) & set(handles.values())
    with pytest.raises(_lib.GtgError):
        owned_stream("cuda", "nonsense")
    # the streams work: a kernel launched on one is ordered against the default stream through events
    x = torch.zeros(1 << 16, device="cuda")
    s = streams["capture"]
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        x.add_(1.0)
    torch.cuda.current_stream().wait_stream(s)
    assert float(x.sum().item()) == float(1 << 16)


def test_graph_capture_beside_a_loader_thread_when_the_pool_would_alias():
    """The failure this guards against: PyTorch's stream pool wraps after 32 draws, so the copy stream a
    prefetching loader draws for an epoch can be the very handle a HIP-graph capture is open on, and the
    loader's thread then issues copies into the capture (hipErrorStreamCaptureUnjoined, or a step graph that
    replays a stale copy).  Here the pool pointer is placed so that the NEXT draw is PyTorch's default capture
    stream, then a trainer captures its step (two shapes: full batches and the ragged tail) while the
    prefetcher's worker pins and copies.  With streams of the package's own and the capture lock the run
    must equal the inline eager run bit for bit, and each shape must be captured exactly once over all
    epochs."""
    from gtg_vae_amd.data import DataLoader, DevicePrefetcher, synth_graph_list
    from gtg_vae_amd.streams import owned_handles
    from gtg_vae_amd.train import train
    from gtg_vae_amd.trainer import GraphVAETrainer

    if torch.cuda.graph.default_capture_stream is None:       # what torch.cuda.graph() does on first use
        torch.cuda.graph.default_capture_stream = torch.cuda.Stream()
    target = int(torch.cuda.graph.default_capture_stream.cuda_stream)
    drawn = _pool_handles(32)
    if target in drawn:
        period = 32
        for _ in range((period - 1 - drawn[::-1].index(target)) % period):   # stop one draw short of `target`
            torch.cuda.Stream()
    N = 12
    data = synth_graph_list(150, N, seed=3)                   # 4 batches of 32 and a tail of 22
    loader = DataLoader(data, batch_size=32, nmax=N)
    runs = []
    for wrap, use_graph in ((False, False), (True, True)):
        tr = GraphVAETrainer(N, 16, 32, N, device="cuda", seed=1)
        ld = DevicePrefetcher(loader, "cuda", depth=3) if wrap else loader
        hist = train(ld, None, tr, epochs=4, generator=torch.Generator().manual_seed(0), quiet=True,
                     use_graph=use_graph)
        assert tr.edges_ok()
        runs.append((tr.flat_p.clone(), [h["loss"] for h in hist], tr.captures))
    assert torch.equal(runs[0][0], runs[1][0]) and runs[0][1] == runs[1][1]
    assert runs[0][2] == 0 and runs[1][2] == 2                # one capture per shape, not per epoch
    handles = owned_handles("cuda")
    assert target not in handles.values()
    assert handles["capture"] != handles["copy"]


def test_loader_thread_waits_while_the_capture_lock_is_held():
    """The prefetcher's worker takes streams.device_work_lock around pin -> copy -> record: while a capture
    (here: the test) holds the lock no batch arrives, and the first one arrives once it is released."""
    import threading

    from gtg_vae_amd.data import DataLoader, DevicePrefetcher, synth_graph_list
    from gtg_vae_amd.streams import device_work_lock

    loader = DataLoader(synth_graph_list(8, 10, seed=1), batch_size=4, nmax=10)
    got = []
    it = iter(DevicePrefetcher(loader, "cuda"))
    with device_work_lock:
        t = threading.Thread(target=lambda: got.append(next(it)), daemon=True)
        t.start()
        t.join(0.5)
        assert t.is_alive() and not got
    t.join(30)
    assert not t.is_alive() and len(got) == 1 and got[0].num_graphs == 4
    assert len(list(it)) == 1


def test_trainer_keeps_one_graph_per_shape_and_survives_an_empty_first_shard():
    """Alternating batch shapes replay the graphs captured for them (no re-capture), results equal eager; a
    rank whose very first shard is empty (run_loaded before any buffer exists) steps without error."""
    from gtg_vae_amd.data import synth_graph_batch
    from gtg_vae_amd.trainer import GraphVAETrainer

    N, d = 10, 16
    big, small = synth_graph_batch(24, N, seed=1).to("cuda"), synth_graph_batch(7, N, seed=2).to("cuda")
    eps = {24: torch.randn(24, d, generator=torch.Generator().manual_seed(0)).cuda(),
           7: torch.randn(7, d, generator=torch.Generator().manual_seed(1)).cuda()}
    runs = []
    for use_graph in (False, True):
        tr = GraphVAETrainer(N, d, 32, N, device="cuda", seed=3, max_batch=24, max_edges=24 * N * 4)
        for i in range(6):
            b = big if i % 2 == 0 else small
            tr.step(b, eps[b.num_graphs], kl_w=0.3, update=True, use_graph=use_graph)
        torch.cuda.synchronize()
        runs.append((tr.flat_p.clone(), tr.captures))
    assert torch.equal(runs[0][0], runs[1][0])
    assert runs[1][1] == 2
    fresh = GraphVAETrainer(N, d, 32, N, device="cuda", seed=3)
    before = fresh.flat_p.clone()
    fresh.load_batch(synth_graph_batch(0, N, seed=1), torch.zeros(0, d))
    res = fresh.run_loaded(update=True)
    torch.cuda.synchronize()
    assert float(res["recon"].item()) == 0.0 and torch.equal(fresh.flat_p, before)   # zero gradients: Adam moves nothing
