#!/usr/bin/env python3
"""Benchmark of the graph-VAE training hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One step = one full training pass over one synthetic batch: adjacency build,
GCN encoder + max-pool, latent head + reparameterisation + KL, MLP decoder,
upper-triangular BCE, the complete backward pass, (all-reduce), Adam.  Inputs
are resident in HBM before the timed region.  Workload = BASELINE.json
configs[2]: N=64 nodes, batch 512 PER GPU (weak scaling), d=512, mlp_hid=512,
fp32.  For N>1 launch with torch.distributed.run (one rank per GPU, RCCL).

On one GPU the timed region replays the step as one HIP graph (--eager: plain
launches); with more GPUs the step is launched eagerly around its two gradient
all-reduces (--graph-segments: three graphs with the collectives between them).
Prints ONE JSON line (rank 0) with the headline metric, the roofline of the
dominant kernel (fp32 MFMA GEMM; executed FLOPs over its HIP-event time, taken
over --profile-steps eager steps that continue the same loop right after the
timed region, which itself carries no events) and, at N=1, the CPU baseline (the
build-owned PyTorch-CPU oracle of the same step, timed on this host's cores)
plus the batched-MST read-out figure.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

PEAK_F32_MFMA_TFLOPS = 157.3   # MI355X dense fp32 matrix peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # MI355X dense bf16 matrix peak
# bf16 MFMA products issued per fp32-equivalent multiply-add in each matrix mode
MODE_PRODUCTS = {"f32": 1, "bf16x3": 6, "f16x2": 3, "bf16": 1}
PEAK_HBM_GBPS = 8000.0         # MI355X HBM3E peak


def flops_per_graph_fwd(N, d, h):
    M = N * (N + 1) // 2
    return 4 * N * N * d + 4 * N * d * d + 2 * N * N * d + 4 * d * d + 2 * d * h + 2 * h * M


def host_cores():
    """Every core this job may use: OMP_NUM_THREADS where the launcher sets it (a one-GPU share of a big
    host), else the process's CPU affinity."""
    env = int(os.environ.get("OMP_NUM_THREADS", "0") or 0)
    if env > 0:
        return env
    return len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)


def cpu_baseline(N, B, d, hid, budget_s=20.0):
    """The oracle's training step (scatter GCN as PyG does it, autograd, Adam)
    on the host cores: >=3 warm-up steps, then timed steps until the budget."""
    from gtg_vae_amd.data import synth_graph_batch
    from oracle import graphvae_oracle as O

    threads = host_cores()
    torch.set_num_threads(threads)
    params = {k: v.clone().requires_grad_(True) for k, v in O.init_params(N, d, hid, N, seed=0).items()}
    opt = torch.optim.Adam(list(params.values()), lr=1e-3, betas=(0.5, 0.999))
    batches = [synth_graph_batch(B, N, seed=1234 + i) for i in range(2)]
    g = torch.Generator().manual_seed(1)
    times = []
    t_start = time.perf_counter()
    i = 0
    while True:
        b = batches[i % 2]
        eps = torch.randn(B, d, generator=g)
        t0 = time.perf_counter()
        opt.zero_grad()
        loss, _, _, _ = O.graph_step_loss(params, None, b.edge_index, b.adj_padded, eps, 0.5, B, N)
        loss.backward()
        opt.step()
        dt = time.perf_counter() - t0
        i += 1
        if i > 3:
            times.append(dt)
        if (i >= 6 and time.perf_counter() - t_start > budget_s) or i >= 40:
            break
    times.sort()
    med = times[len(times) // 2]
    return {"value": B / med, "unit": "graphs/s", "cores": threads, "kind": "port",
            "sample": "%d timed steps of the same workload (B=%d, N=%d, d=%d), median step %.1f ms, on every core this "
                      "job may use; oracle = eager PyTorch-CPU scatter-GCN step with autograd + Adam" % (
                          len(times), B, N, d, med * 1e3)}


def mst_cpu_baseline(rows_host, N):
    """The C restatement of maxspantree.py (oracle/mst_oracle.c) on one host core over a bounded sample of the
    same rows, checked against nothing here (the parity tests do that): a reported baseline only."""
    from oracle import c_mst

    c_mst.mst_triu(rows_host[:64], N)  # warm
    n = 4096
    t0 = time.perf_counter()
    c_mst.mst_triu(rows_host[:n], N)
    dt = time.perf_counter() - t0
    res = {"value": n / dt, "unit": "graphs/s", "cores": 1, "kind": "port",
           "sample": "%d of the bench's decoder rows through the C restatement of maxspantree.py, one core, %.1f ms "
                     "(the reference's own Python takes ~61 ms per graph at N=64)" % (n, dt * 1e3)}
    # the same code on every core this process may use: graphs are independent, ctypes releases the GIL
    from concurrent.futures import ThreadPoolExecutor
    cores = host_cores()
    per = 2048
    total = min(len(rows_host), cores * per)
    chunks = [rows_host[i:i + per] for i in range(0, total, per)]
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        list(ex.map(lambda c: c_mst.mst_triu(c, N), chunks))
        dt = time.perf_counter() - t0
    res["all_cores"] = {"value": total / dt, "unit": "graphs/s", "cores": cores,
                        "sample": "%d rows in chunks of %d over a pool of %d threads, %.1f ms" % (total, per, cores, dt * 1e3)}
    return res


PMC_FILE = "profiles/r03_pmc_traffic.json"
PMC_SOURCE_NOTE = ("HBM bytes per launch (2 x FETCH_SIZE + WRITE_SIZE) from separate rocprofv3 --pmc passes over this "
                   "launch shape, committed as %s; NOT measured in this run (null: no pass on record)" % PMC_FILE)


def pmc_traffic(key):
    try:
        with open(os.path.join(ROOT, PMC_FILE)) as f:
            return json.load(f).get(key, {}).get("traffic_bytes")
    except OSError:
        return None


def small_config_bench(dev, N, B, d, hid, steps=100):
    """A short run of another named configuration (HIP-graph replay, resident inputs), for the sub-records."""
    from gtg_vae_amd.data import synth_graph_batch
    from gtg_vae_amd.trainer import GraphVAETrainer

    tr = GraphVAETrainer(N, d, hid, N, device=dev, seed=0, max_batch=B, max_edges=B * N * 4)
    batches = [synth_graph_batch(B, N, seed=77 + i).to(dev) for i in range(2)]
    eps = [torch.randn(B, d, generator=torch.Generator().manual_seed(i)).to(dev) for i in range(2)]
    tr.set_kl_weight(0.5)
    for i in range(10):
        tr.load_batch(batches[i % 2], eps[i % 2])
        tr.run_loaded(update=True, use_graph=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(steps):
        tr.load_batch(batches[i % 2], eps[i % 2])
        tr.run_loaded(update=True, use_graph=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"workload": "synthetic graphs N<=%d, batch %d, d=%d, mlp_hid=%d, graph-only step" % (N, B, d, hid),
            "ms_per_step": dt * 1e3, "graphs_per_s": B / dt}


def joint_bench(dev, steps=8):
    """BASELINE configs[4] on one GPU: the joint graph+text step through JointTrainer (batch 256, N=32, d=256,
    sentences of 8..40 tokens, vocabulary 30 522, dropout and word noise on)."""
    from gtg_vae_amd.data import collate, synth_graph_list
    from gtg_vae_amd.joint_trainer import JointTrainer

    B, N, d, hid, V, T = 256, 32, 256, 512, 30522, 40
    gb = collate(synth_graph_list(B, N, seed=1234, vocab=V, max_len=T), N).to(dev)
    eps = torch.randn(B, d, generator=torch.Generator().manual_seed(0)).to(dev)
    tr = JointTrainer(N, d, hid, N, V, 0, device=dev, seed=0)
    tr.model.train()
    for _ in range(3):
        tr.step(gb, gb.dec_seq, eps, 0.3, update=True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        tr.step(gb, gb.dec_seq, eps, 0.3, update=True)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    return {"workload": "configs[4] on one GPU: joint graph+text step, batch 256, N=32, d=256, T<=40, vocab 30522; "
                        "HIP graph half + HIP LSTM recurrences + HIP vocabulary projection/cross-entropy + flat fused Adam",
            "matrix_mode": "f16x2 (the library default: fp32 operands and results, two scaled fp16 planes and three "
                           "products per multiply-add in the large products and the LSTM recurrence)",
            "ms_per_step": dt * 1e3, "graphs_per_s": B / dt, "parameters": tr.numel}


def mst_bench(device, G=65536, N=64, reps=5, cpu=True):
    """BASELINE.json configs[3]: 65 536 decoder rows (N=64) -> spanning trees."""
    from gtg_vae_amd import _lib, mst_graph_batched

    M = N * (N + 1) // 2
    g = torch.Generator().manual_seed(99)
    rows = torch.rand((G, M), generator=g)
    iu = torch.triu_indices(N, N)
    dpos = torch.nonzero(iu[0] == iu[1]).flatten()
    rows[:, dpos] = 0.3 + 0.7 * torch.rand((G, N), generator=g)
    rows_host = rows.numpy()
    rows = rows.to(device)
    mst_graph_batched(rows, N)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for _ in range(reps):
        mst_graph_batched(rows, N)
    torch.cuda.synchronize()
    rep = [r for r in _lib.profile_report() if r[0] == "mst_triu"]
    _lib.profile_enable(False)
    ms = rep[0][5] / rep[0][4]
    bytes_per_graph = 4 * M + 4 * (2 * (N - 1) + 1)
    gbps = G * bytes_per_graph / (ms * 1e-3) / 1e9
    traffic = pmc_traffic("mst_triu %d %d" % (G, N))
    res = {"workload": "configs[3]: %d graphs, N=%d, upper-triangular fp32 rows" % (G, N),
           "graphs_per_s": G / (ms * 1e-3), "ms": ms,
           "roofline": {"bound": "hbm", "achieved": gbps, "peak": PEAK_HBM_GBPS, "unit": "GB/s",
                        "frac": gbps / PEAK_HBM_GBPS, "traffic": traffic, "traffic_source": PMC_SOURCE_NOTE,
                        "algorithmic_bytes_per_graph": bytes_per_graph}}
    if cpu:
        res["cpu_baseline"] = mst_cpu_baseline(rows_host, N)
    # the same read-out at the reference's default size (N = 10, train.py:544: four graphs share a wavefront) and
    # at N = 32, same number of graphs, same distribution
    other = {}
    for n2 in (10, 32):
        m2 = n2 * (n2 + 1) // 2
        r2 = torch.rand((G, m2), generator=g)
        iu2 = torch.triu_indices(n2, n2)
        r2[:, torch.nonzero(iu2[0] == iu2[1]).flatten()] = 0.3 + 0.7 * torch.rand((G, n2), generator=g)
        r2 = r2.to(device)
        mst_graph_batched(r2, n2)
        torch.cuda.synchronize()
        _lib.profile_enable(True)
        for _ in range(reps):
            mst_graph_batched(r2, n2)
        torch.cuda.synchronize()
        rep2 = [r for r in _lib.profile_report() if r[0] == "mst_triu"]
        _lib.profile_enable(False)
        ms2 = rep2[0][5] / rep2[0][4]
        b2 = 4 * m2 + 4 * (2 * (n2 - 1) + 1)
        other["N=%d" % n2] = {"ms": ms2, "graphs_per_s": G / (ms2 * 1e-3), "GB_per_s": G * b2 / (ms2 * 1e-3) / 1e9,
                              "algorithmic_bytes_per_graph": b2}
    res["other_sizes"] = other
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--repeats", type=int, default=5,
                    help="the --steps-step region is timed this many times back to back; the median region is reported")
    ap.add_argument("--nodes", type=int, default=64)
    ap.add_argument("--batch", type=int, default=512, help="graphs per GPU per step")
    ap.add_argument("--latent", type=int, default=512)
    ap.add_argument("--hid", type=int, default=512)
    ap.add_argument("--strong", action="store_true",
                    help="strong scaling: --batch is the global batch, split evenly over the GPUs (default: weak, --batch per GPU)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-mst", action="store_true")
    ap.add_argument("--no-extra", action="store_true", help="skip the sub-records of configurations A, B and the joint step")
    ap.add_argument("--matrix", choices=["f32", "bf16x3", "f16x2", "bf16"], default="f16x2",
                    help="matrix pipe of the large encoder GEMMs: fp32 MFMA; the exact three-way bf16 split on the bf16 "
                         "MFMA; the scaled two-way fp16 split on the fp16 MFMA (both fp32-level error; f16x2 is the "
                         "default); plain bf16 operands (loss tolerance 1e-3)")
    ap.add_argument("--no-alt", action="store_true", help="skip the short extra runs in the other matrix modes")
    ap.add_argument("--eager", action="store_true", help="launch every kernel from the host instead of replaying HIP graphs")
    ap.add_argument("--graph-segments", action="store_true",
                    help="with more than one GPU: replay the launches between the all-reduces as HIP graphs (default there: eager)")
    ap.add_argument("--profile-steps", type=int, default=20, help="eager steps after the timed region, for per-kernel HIP-event times")
    args = ap.parse_args()
    if os.environ.get("GTG_HANG_DUMP"):  # debugging aid: dump all Python stacks if the run stalls
        import faulthandler

        faulthandler.dump_traceback_later(int(os.environ["GTG_HANG_DUMP"]), exit=True)

    from gtg_vae_amd import _lib
    from gtg_vae_amd.data import synth_graph_batch
    from gtg_vae_amd.parallel import init_process_group, local_device_index
    from gtg_vae_amd.trainer import GraphVAETrainer

    rank, local_rank, world = init_process_group()
    if world != args.gpus:
        if rank == 0:
            print("warning: --gpus %d but WORLD_SIZE=%d; using WORLD_SIZE" % (args.gpus, world), file=sys.stderr)
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the hot path has no CPU fallback")
    dev = torch.device("cuda", local_device_index(local_rank))
    torch.cuda.set_device(dev)
    N, B, d, hid = args.nodes, args.batch, args.latent, args.hid
    if args.strong:
        if B % world:
            raise SystemExit("--strong needs --batch divisible by the number of GPUs")
        B //= world

    # identical initial weights on every rank, different data per rank
    tr = GraphVAETrainer(N, d, hid, N, device=dev, seed=0, max_batch=B, max_edges=B * N * 4, matrix_mode=args.matrix)
    nb = 4
    batches = [synth_graph_batch(B, N, seed=1234 + rank * 1000 + i).to(dev) for i in range(nb)]
    gen = torch.Generator().manual_seed(4321 + rank)
    eps = [torch.randn(B, d, generator=gen).to(dev) for _ in range(nb)]
    tr.set_kl_weight(0.5)
    # One GPU: the step is replayed as a single HIP graph.  More than one: the launches between the two gradient
    # all-reduces could be replayed as three graphs, but with a live RCCL group the eagerly launched step is the
    # faster of the two (tools/rccl_graph_probe.py: 0.604 against 0.622 ms) -- the host runs far enough ahead of the
    # GPU for the launches to cost nothing, and a collective between two graph launches costs more than between
    # two kernels.  --graph-segments replays the segments as graphs all the same.
    use_graph = not args.eager and (world == 1 or args.graph_segments)

    def one_step(i, graph=use_graph):
        tr.load_batch(batches[i % nb], eps[i % nb])
        tr.run_loaded(update=True, use_graph=graph)

    # census of the encoder's stored rows (R real + N prototype rows per batch) and per-graph extents: the launches
    # tagged ":m"/":k"/":g" read these on the device, so their flop counts below use what was executed
    census = []
    for i in range(nb):
        tr.load_batch(batches[i], eps[i])
        tr.run_loaded(update=False)
        torch.cuda.synchronize()
        meta = tr.meta[: 2 * B + 2].cpu()
        nb_real = meta[:B].double()
        census.append({"rows": float(meta[2 * B + 1]), "sq": float((nb_real * nb_real).sum()), "sum": float(nb_real.sum())})
    used = [(args.warmup + i) % nb for i in range(args.steps)]
    eff_rows = sum(census[j]["rows"] for j in used) / max(1, len(used))
    eff_sq = sum(census[j]["sq"] for j in used) / max(1, len(used))

    for i in range(args.warmup):
        one_step(i)
    torch.cuda.synchronize()

    def timed_region(first):
        """EXACTLY args.steps steps between two (barrier + device synchronize) brackets; the slowest rank's time."""
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(args.steps):
            one_step(first + i)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if world > 1:
            from gtg_vae_amd.parallel import all_reduce_small

            dt = all_reduce_small(torch.tensor([dt], dtype=torch.float64, device=dev), dist.ReduceOp.MAX).item()
        return dt

    # the region is timed --repeats times back to back (the loop simply continues) and the MEDIAN region is the
    # one reported: a single 12 ms sample is at the mercy of one stray host stall
    repeats = max(1, args.repeats)
    regions = [timed_region(args.warmup + r * args.steps) for r in range(repeats)]
    elapsed = sorted(regions)[len(regions) // 2]
    steps_done = args.warmup + repeats * args.steps
    # The timed region carries no events (a replayed graph has no per-launch events anyway): the same steps are
    # continued eagerly, outside it, with HIP events around every GEMM launch (same kernels, shapes and batches;
    # every rank runs them, so that the collectives still match).
    prof_steps = max(1, args.profile_steps)
    _lib.profile_enable(True)
    used = [(steps_done + i) % nb for i in range(prof_steps)]
    eff_rows = sum(census[j]["rows"] for j in used) / len(used)
    eff_sq = sum(census[j]["sq"] for j in used) / len(used)
    for i in range(prof_steps):
        one_step(steps_done + i, graph=False)
    torch.cuda.synchronize()
    prof = _lib.profile_report()
    _lib.profile_enable(False)
    # kernel launches of one step (the library counts its launches on the host)
    c0 = _lib.lib().gtg_launch_count()
    one_step(0, graph=False)
    torch.cuda.synchronize()
    launches_per_step = int(_lib.lib().gtg_launch_count() - c0)
    if not tr.edges_ok():
        raise SystemExit("adjacency kernel flagged an invalid edge list")
    recon, kl = tr.scalars[0].item(), tr.scalars[1].item()

    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * B * args.steps / elapsed
        F = flops_per_graph_fwd(N, d, hid)
        # dominant kernel: the 128x128-tile fp32 MFMA GEMM launch with the largest total time
        roof = None
        kern = {}
        big = [r for r in prof if r[0].startswith("gemm")]

        def executed_flops(r):
            name, m_, n_, k_ = r[:4]
            if name.endswith(":m"):
                return 2.0 * eff_rows * n_ * k_
            if name.endswith(":k"):
                return 2.0 * m_ * n_ * eff_rows
            if name.endswith(":g"):     # per graph (n_b x N x n_b)
                return 2.0 * eff_sq * n_
            return 2.0 * m_ * n_ * k_

        if big:
            top = max(big, key=lambda r: r[5])
            name, m_, n_, k_, cnt, tot_ms = top
            fl = executed_flops(top)
            avg_ms = tot_ms / cnt
            ach = fl / (avg_ms * 1e-3) / 1e12
            # HBM bytes per launch come from separate rocprofv3 --pmc passes (bench.py cannot read PMCs itself)
            traffic = pmc_traffic("%s %s %d %d %d" % (args.matrix, name, m_, n_, k_))
            # the 128x128 GEMMs run on the pipe --matrix selects: price them against that pipe's peak, counting the
            # MFMA products actually issued (six per multiply-add for the three-way split)
            on_bf16 = args.matrix != "f32" and name.startswith("gemm128")
            peak = PEAK_BF16_MFMA_TFLOPS if on_bf16 else PEAK_F32_MFMA_TFLOPS
            products = MODE_PRODUCTS[args.matrix]
            issued = ach * (products if on_bf16 else 1)
            roof = {"bound": "mfma", "achieved": issued, "peak": peak, "unit": "TFLOP/s",
                    "frac": issued / peak, "traffic": traffic, "traffic_source": PMC_SOURCE_NOTE, "kernel": name,
                    "pipe": ("v_mfma_f32_32x32x16_bf16 / _f16, %d products per fp32 multiply-add" % products)
                            if on_bf16 else "v_mfma_f32_32x32x2_f32",
                    "fp32_equivalent_tflops": ach,
                    # the same launch priced as what it computes (fp32 multiply-adds) against the fp32 matrix peak
                    "fp32_equivalent_frac_of_f32_mfma_peak": ach / PEAK_F32_MFMA_TFLOPS,
                    "launch": {"M": m_, "N": n_, "K": k_, "flops": fl, "avg_ms": avg_ms, "launches": cnt,
                               "stored_rows": eff_rows if ":" in name else None,
                               "timed_with": "HIP events on the launch stream, %d eager steps right after the timed %s" % (
                                   prof_steps, "graph replays" if use_graph else "steps")}}
            for r in sorted(big, key=lambda r: -r[5])[:8]:
                kern["%s M%d N%d K%d" % r[:4]] = {"avg_ms": r[5] / r[4], "tflops": executed_flops(r) / (r[5] / r[4] * 1e-3) / 1e12}
        # FLOPs the step's GEMM launches executed (the sparse row-product kernels of the encoder, which replace
        # dense products of the padded formula, and the element-wise kernels are not counted), per step and GPU
        gemm_flops = sum(executed_flops(r) * r[4] for r in big) / prof_steps
        line = {
            "metric": ("training graphs/sec (whole node) at N=64 nodes, batch 512; 1/2/4/8 MI355X"
                       if (N, (world * B if args.strong else B), d, hid) == (64, 512, 512, 512) else
                       "training graphs/sec (whole node) at N=%d nodes, batch %d (not the headline configuration)" % (N, B)),
            "value": value, "unit": "graphs/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "repeats": {"n": repeats, "reported": "median region", "ms_per_step_min": min(regions) / args.steps * 1e3,
                        "ms_per_step_max": max(regions) / args.steps * 1e3,
                        "ms_per_step_all": [r / args.steps * 1e3 for r in regions]},
            "higher_is_better": True, "scaling": "strong" if args.strong else "weak", "vs_baseline": None,
            "dtype": "bf16" if args.matrix == "bf16" else "f32", "data": "synthetic",
            "config": {"workload": "%ssynthetic AMR-shaped graphs N<=%d (padded to %d), batch %d per GPU, "
                                   "latent d=%d, mlp_hid=%d, graph-only step: recon + kl_w*KL, full backward, Adam" % (
                                       "BASELINE configs[2]: " if (N, B, d, hid) == (64, 512, 512, 512) else "", N, N, B, d, hid),
                       "global_batch": world * B, "parallelism": "dp%d" % world, "hip_graph": bool(use_graph),
                       "matrix_pipe": {"f32": "fp32 MFMA everywhere",
                                       "bf16x3": "fp32 operands and results; the large encoder GEMMs split every operand "
                                                 "exactly into three bf16 pieces and sum six piece products on the bf16 MFMA "
                                                 "with fp32 accumulation (error vs float64 equal to the fp32 MFMA's; the "
                                                 "parity tests run at the fp32 tolerances in this mode)",
                                       "f16x2": "fp32 operands and results; the large encoder GEMMs scale every operand (row) "
                                                "by a power of two, split it into two fp16 pieces (round to nearest twice) "
                                                "and sum three piece products on the fp16 MFMA with fp32 accumulation, the "
                                                "result scaled back exactly (error vs float64 equal to the fp32 MFMA's; the "
                                                "parity tests run at the fp32 tolerances in this mode)",
                                       "bf16": "large encoder GEMMs on bf16-rounded operands, fp32 accumulation "
                                               "(loss tolerance 1e-3)"}[args.matrix]},
            "step_executed": {"gemm_gflop_per_step": gemm_flops / 1e9,
                              "gemm_tflops_over_whole_step": gemm_flops / (ms_per_step * 1e-3) / 1e12,
                              "frac_of_f32_mfma_peak": gemm_flops / (ms_per_step * 1e-3) / 1e12 / PEAK_F32_MFMA_TFLOPS,
                              "padded_formula_gflop_per_step": 3 * F * B / 1e9,
                              "launches_per_step": launches_per_step,
                              "note": "executed multiply-adds of the GEMM launches only, as fp32 FLOPs, over the step time; "
                                      "the padded formula counts rows and dense products the compact-row encoder skips"},
            "encoder_rows": {"padded": B * N, "stored_mean": eff_rows},
            "loss": {"recon": recon, "kl": kl},
            "roofline": roof,
            "gemm_kernels": kern,
        }
        if world == 1 and not args.no_alt:
            # the same loop in the other matrix modes (short, after the headline measurement)
            alt = {}
            for mode in ("f32", "bf16x3", "f16x2", "bf16"):
                if mode == args.matrix:
                    continue
                tr.matrix_mode = _lib.matrix_mode(mode)
                tr.drop_graphs()
                for i in range(5):
                    one_step(i)
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                for i in range(30):
                    one_step(i)
                torch.cuda.synchronize()
                dt = (time.perf_counter() - t1) / 30
                alt[mode] = {"ms_per_step": dt * 1e3, "graphs_per_s": B / dt}
            tr.matrix_mode = _lib.matrix_mode(args.matrix)
            tr.drop_graphs()
            line["other_matrix_modes"] = alt
            # the same step with every product on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32), for readers
            # who do not accept split-fp16 arithmetic under the label "f32"
            strict = alt.get("f32") or {"ms_per_step": ms_per_step, "graphs_per_s": value}
            line["value_f32_mfma_only"] = strict["graphs_per_s"]
            line["ms_per_step_f32_mfma_only"] = strict["ms_per_step"]
        if world == 1 and not args.no_mst:
            line["mst"] = mst_bench(dev, cpu=not args.no_cpu_baseline)
        if world == 1 and not args.no_extra:
            del tr
            torch.cuda.empty_cache()
            line["config_b"] = small_config_bench(dev, 32, 256, 256, 512)
            line["config_a"] = small_config_bench(dev, 10, 64, 32, 512)
            line["joint"] = joint_bench(dev)
        if world == 1 and not args.no_cpu_baseline:
            cb = cpu_baseline(N, B, d, hid)
            line["cpu_baseline"] = cb
            line["speedup_vs_cpu_baseline"] = value / cb["value"]
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
