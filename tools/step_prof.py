"""Per-launch-site times of the graph step (HIP events of the library's own profiler around every tagged launch),
eager steps of the headline configuration, and a checksum of every gradient after one fixed step so that two
builds can be compared bit for bit:  python tools/step_prof.py [--steps 40] [--tags dw2_sparse,dw1_sparse]"""
import argparse
import hashlib
import json
import os
import sys

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--nodes", type=int, default=64)
    ap.add_argument("--batch", type=int, default=512)
    ap.add_argument("--latent", type=int, default=512)
    ap.add_argument("--hid", type=int, default=512)
    ap.add_argument("--matrix", default="f16x2")
    ap.add_argument("--tags", default="")
    args = ap.parse_args()
    from gtg_vae_amd import _lib
    from gtg_vae_amd.data import synth_graph_batch
    from gtg_vae_amd.trainer import GraphVAETrainer

    dev = torch.device("cuda", 0)
    N, B, d, hid = args.nodes, args.batch, args.latent, args.hid
    tr = GraphVAETrainer(N, d, hid, N, device=dev, seed=0, max_batch=B, max_edges=B * N * 4, matrix_mode=args.matrix)
    batches = [synth_graph_batch(B, N, seed=1234 + i).to(dev) for i in range(4)]
    gen = torch.Generator().manual_seed(4321)
    eps = [torch.randn(B, d, generator=gen).to(dev) for _ in range(4)]
    tr.set_kl_weight(0.5)
    # one step without the update: gradients of fixed weights on a fixed batch
    tr.load_batch(batches[0], eps[0])
    tr.run_loaded(update=False, use_graph=False)
    torch.cuda.synchronize()
    sums = {}
    for name, g in tr.named_grads().items():
        sums[name] = hashlib.sha1(g.detach().cpu().numpy().tobytes()).hexdigest()[:12]
    for i in range(10):
        tr.load_batch(batches[i % 4], eps[i % 4])
        tr.run_loaded(update=True, use_graph=False)
    torch.cuda.synchronize()
    _lib.profile_enable(True)
    for i in range(args.steps):
        tr.load_batch(batches[i % 4], eps[i % 4])
        tr.run_loaded(update=True, use_graph=False)
    torch.cuda.synchronize()
    rep = _lib.profile_report()
    _lib.profile_enable(False)
    want = [t for t in args.tags.split(",") if t]
    out = {"grad_sha1": sums, "us_per_launch": {}}
    for r in sorted(rep, key=lambda r: -r[5]):
        if want and not any(r[0].startswith(t) for t in want):
            continue
        out["us_per_launch"]["%s %d %d %d" % r[:4]] = round(r[5] / r[4] * 1e3, 2)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
