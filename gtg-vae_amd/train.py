"""Training / evaluation loop with the reference's structure and printed fields.

Mirrors ``train()`` and ``dev()`` of /root/reference/src/train.py:284-355 and
:228-282: Adam(lr=1e-3, betas=(0.5, 0.999)), KL weight annealed per EPOCH as
min(epoch / 500, 1) (train.py:294), batch-size-weighted epoch means, the same
printed lines (train.py:343-352, 273-279), a dev pass after every epoch.  The
step itself is GraphVAETrainer's fused HIP path; with several ranks
(torchrun) every rank trains on its own shard of each batch and gradients are
all-reduced over RCCL.

    python -m gtg_vae_amd.train --synthetic 4096 --nodes 10 --batch 64 --latent 32 --epochs 3
    python -m gtg_vae_amd.train --train amr.mrp --dev dev.mrp --nodes 10
"""
import argparse
import time

import torch

from .data import DataLoader, load_mrp, prep_data_list, synth_graph_list
from .parallel import init_process_group, local_device_index, shard_range
from .trainer import GraphVAETrainer

NKL = 500  # epochs over which the KL weight ramps to 1 (train.py:286,294)


def draw_eps(B: int, d: int, generator, dev) -> torch.Tensor:
    """The latent noise of one batch (gtvae.py:36 draws it with randn_like on the device): drawn on the device,
    from ``generator`` when it is a generator of that device or None; a CPU generator draws on the host and
    copies (reproducible across loaders and devices, which the tests use -- and a quarter of a millisecond of
    host time per 512 x 512 batch, more than a third of the step)."""
    if generator is None or generator.device.type == "cuda":
        return torch.randn(B, d, device=dev, generator=generator)
    return torch.randn(B, d, generator=generator).to(dev)


def dev(loader, trainer: GraphVAETrainer, epoch, kl_anneal_w, generator=None, quiet=False):
    sums = torch.zeros(5, dtype=torch.float64)
    n_inst = 0
    for batch in loader:
        B = batch.num_graphs
        eps = draw_eps(B, trainer.d, generator, trainer.dev)
        res = trainer.evaluate(batch.to(trainer.dev), eps, report=True)
        vals = torch.stack([res["recon"] + kl_anneal_w * res["kl"], res["recon"], res["kl"], res["edge_recall"],
                            res["edge_precision"]]).double().cpu()
        sums += vals * B
        n_inst += B
    means = (sums / max(n_inst, 1)).tolist()
    if not quiet:
        print("DEV")
        print("Epoch: {}".format(epoch))
        print("Epoch loss: {:.3f}".format(means[0]))
        print("Epoch graph_recon loss: {:.3f}".format(means[1]), "Epoch text_recon_loss: {:.3f}".format(0.0))
        print("Epoch joint_kl_loss: {:.3f}".format(means[2]))
        print("Epoch edge_recall: {:.3f}".format(means[3]), "Epoch edge_precision: {:.3f}".format(means[4]))
        print("Epoch text_acc: {:.3f}".format(0.0))
    return dict(zip(("loss", "graph_recon", "kl", "edge_recall", "edge_precision"), means))


def train(train_loader, dev_loader, trainer: GraphVAETrainer, epochs, model_name="gtg", rank=0, world=1,
          generator=None, use_graph=False, quiet=False, save_every=0):
    lr = trainer.lr
    if rank == 0 and not quiet:
        print("trnsize: {}, devsize: {}, num_epochs: {}, nkl: {}, at: {}, lr: {}".format(
            len(train_loader.dataset), len(dev_loader.dataset) if dev_loader else 0, epochs, NKL, 0.3, lr))
    history = []
    for epoch in range(epochs):
        kl_anneal_w = min(epoch / NKL, 1)
        trainer.set_kl_weight(kl_anneal_w)
        sums = torch.zeros(3, dtype=torch.float64, device=trainer.dev)
        n_inst = 0
        t0 = time.perf_counter()
        for batch in train_loader:
            if world > 1 and not getattr(train_loader, "sharded", False):
                lo, hi = shard_range(batch.num_graphs, rank, world)
                batch = slice_batch(batch, lo, hi)
            if isinstance(batch, torch.Tensor):   # graph ids of a dataset resident on the device
                B = int(batch.numel())
                eps = draw_eps(B, trainer.d, generator, trainer.dev)
                trainer.load_indices(train_loader.dataset, batch, eps)
            else:
                B = batch.num_graphs
                eps = draw_eps(B, trainer.d, generator, trainer.dev)
                trainer.load_batch(batch if batch.edge_index.is_cuda else batch.to(trainer.dev), eps)
            res = trainer.run_loaded(update=True, use_graph=use_graph)
            # device-side accumulation: no host sync inside the epoch (the reference syncs with loss.item(), train.py:316)
            sums += torch.stack([res["recon"] + kl_anneal_w * res["kl"], res["recon"], res["kl"]]).double() * B
            n_inst += B
        torch.cuda.synchronize(trainer.dev)
        dt = time.perf_counter() - t0
        if not trainer.edges_ok():
            raise RuntimeError("epoch %d: an edge left its graph or a gathered batch overflowed the edge buffer" % epoch)
        tot = torch.cat([sums, torch.tensor([float(n_inst)], dtype=torch.float64, device=trainer.dev)])
        if world > 1:
            from .parallel import all_reduce_small

            tot = all_reduce_small(tot)
        means = (tot[:3] / tot[3]).tolist()
        history.append({"epoch": epoch, "loss": means[0], "graph_recon": means[1], "kl": means[2],
                        "graphs_per_s": tot[3].item() / dt})
        if rank == 0 and not quiet:
            print("\n ============================")
            print("Epoch: {}".format(epoch))
            print("Epoch loss: {:.3f}".format(means[0]))
            print("Epoch graph_recon loss: {:.3f}".format(means[1]), "Epoch text_recon_loss: {:.3f}".format(0.0))
            print("Epoch joint_kl_loss: {:.3f}".format(means[2]))
            print("kl_anneal_w:", kl_anneal_w)
            print("graphs/s: {:.0f}".format(tot[3].item() / dt))
        if dev_loader is not None and rank == 0:
            history[-1]["dev"] = dev(dev_loader, trainer, epoch, kl_anneal_w, generator, quiet)
        if save_every and rank == 0 and epoch % save_every == 0:
            torch.save(trainer.state_dict(), "%s_%d" % (model_name, epoch))  # cadence of train.py:356-358
    return history


def _print_epoch(tag, epoch, m, kl_anneal_w=None, graphs_per_s=None):
    """The reference's epoch report (train.py:343-352, dev: 273-279), same field names and formats."""
    print("DEV" if tag == "dev" else "\n ============================")
    print("Epoch: {}".format(epoch))
    print("Epoch loss: {:.3f}".format(m["loss"]))
    print("Epoch graph_recon loss: {:.3f}".format(m["graph_recon"]), "Epoch text_recon_loss: {:.3f}".format(m["text_recon"]))
    print("Epoch joint_kl_loss: {:.3f}".format(m["kl"]))
    print("Epoch edge_recall: {:.3f}".format(m["edge_recall"]), "Epoch edge_precision: {:.3f}".format(m["edge_precision"]))
    print("Epoch text_acc: {:.3f}".format(m["text_acc"]))
    if kl_anneal_w is not None:
        print("kl_anneal_w:", kl_anneal_w)
    if graphs_per_s is not None:
        print("graphs/s: {:.0f}".format(graphs_per_s))


_JOINT_FIELDS = ("loss", "text_acc", "kl", "graph_recon", "text_recon", "edge_recall", "edge_precision")


def dev_joint(loader, trainer, epoch, kl_anneal_w, generator=None, quiet=False):
    """dev() of train.py:228-282 for the joint model: eval mode (no dropout, no word noise; the latent is
    still sampled, gtvae.py:34), batch-size-weighted means, spanning-tree edge recall / precision reported."""
    was_training = trainer.model.training
    trainer.model.eval()
    sums = torch.zeros(len(_JOINT_FIELDS), dtype=torch.float64)
    n_inst = 0
    for batch in loader:
        B = batch.num_graphs
        eps = draw_eps(B, trainer.d, generator, trainer.dev)
        gb = batch.to(trainer.dev)
        res = trainer.evaluate(gb, gb.dec_seq, eps, kl_anneal_w, report=True)
        sums += torch.stack([torch.as_tensor(res[k], dtype=torch.float64).cpu() for k in _JOINT_FIELDS]) * B
        n_inst += B
    trainer.model.train(was_training)
    means = dict(zip(_JOINT_FIELDS, (sums / max(n_inst, 1)).tolist()))
    if not quiet:
        _print_epoch("dev", epoch, means)
    return means


def train_joint(train_loader, dev_loader, trainer, epochs, model_name="gtg", rank=0, world=1, generator=None,
                quiet=False, save_every=0):
    """train() of train.py:284-355 for the joint graph+text model: model.train() (dropout 0.5 and word drop
    0.3 on), KL weight min(epoch/500, 1) per epoch, one JointTrainer.step per batch (every rank on its shard of
    it), batch-size-weighted epoch means accumulated on the device, the reference's printed fields."""
    if rank == 0 and not quiet:
        print("trnsize: {}, devsize: {}, num_epochs: {}, nkl: {}, at: {}, lr: {}".format(
            len(train_loader.dataset), len(dev_loader.dataset) if dev_loader else 0, epochs, NKL, 0.3, trainer.lr))
    fields = _JOINT_FIELDS[:5]
    history = []
    for epoch in range(epochs):
        trainer.model.train()
        kl_anneal_w = min(epoch / NKL, 1)
        sums = torch.zeros(len(fields), dtype=torch.float64, device=trainer.dev)
        n_inst = 0
        t0 = time.perf_counter()
        for batch in train_loader:
            if world > 1 and not getattr(train_loader, "sharded", False):
                lo, hi = shard_range(batch.num_graphs, rank, world)
                batch = slice_batch(batch, lo, hi)
            B = batch.num_graphs
            eps = draw_eps(B, trainer.d, generator, trainer.dev)
            gb = batch if batch.edge_index.is_cuda else batch.to(trainer.dev)
            res = trainer.step(gb, gb.dec_seq, eps, kl_anneal_w, update=True)
            sums += torch.stack([res[k].double() for k in fields]) * B
            n_inst += B
        torch.cuda.synchronize(trainer.dev)
        dt = time.perf_counter() - t0
        tot = torch.cat([sums, torch.tensor([float(n_inst)], dtype=torch.float64, device=trainer.dev)])
        if world > 1:
            from .parallel import all_reduce_small

            tot = all_reduce_small(tot)
        means = dict(zip(fields, (tot[:-1] / tot[-1]).tolist()))
        means.update(edge_recall=0.0, edge_precision=0.0)   # training passes report=False (gtvae.py:123)
        history.append(dict(means, epoch=epoch, graphs_per_s=tot[-1].item() / dt))
        if rank == 0 and not quiet:
            _print_epoch("train", epoch, means, kl_anneal_w, tot[-1].item() / dt)
        if dev_loader is not None and rank == 0:
            history[-1]["dev"] = dev_joint(dev_loader, trainer, epoch, kl_anneal_w, generator, quiet)
        if save_every and rank == 0 and epoch % save_every == 0:
            torch.save(trainer.state_dict(), "%s_%d" % (model_name, epoch))
    return history


class ShardedLoader:
    """This rank's contiguous share of every batch of ``loader`` (host side, before any copy)."""
    sharded = True

    def __init__(self, loader, rank, world):
        self.loader, self.rank, self.world = loader, rank, world
        self.dataset = getattr(loader, "dataset", None)

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for batch in self.loader:
            lo, hi = shard_range(batch.num_graphs, self.rank, self.world)
            yield slice_batch(batch, lo, hi)


def slice_batch(batch, lo, hi):
    """Graphs [lo, hi) of a collated batch (host tensors), renumbered from 0 and tagged with the size of the
    batch they were cut from (the trainer weights an unequal shard's gradient by it); hi == lo is legal."""
    from .data import GraphBatch

    n = batch.nmax
    ei = batch.edge_index
    g = ei[1] // n
    keep = (g >= lo) & (g < hi)
    ei = (ei[:, keep] - lo * n).contiguous()
    B = hi - lo
    return GraphBatch(ei, batch.batch[: B * n].clone(), batch.adj_padded[lo:hi].contiguous(), batch.gold_edges[lo:hi],
                      batch.snts[lo:hi], B, n, batch.num_nodes[lo:hi], None, batch.gold_bits[lo:hi].contiguous(),
                      batch.num_graphs,
                      # all T columns are kept: the text encoder reads the padding too (text_vae.py:56 runs the LSTM
                      # over the padded batch), so a shard must see the sentence length of the whole batch
                      None if batch.dec_seq is None else batch.dec_seq[lo:hi].contiguous())


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__)
    ap.add_argument("--train", help="MRP JSON-lines file (nodes / edges[{source,target}] / input)")
    ap.add_argument("--dev")
    ap.add_argument("--synthetic", type=int, default=0, help="number of synthetic AMR-shaped training graphs")
    ap.add_argument("--nodes", type=int, default=10)
    ap.add_argument("--batch", type=int, default=64)
    ap.add_argument("--latent", type=int, default=32)
    ap.add_argument("--hid", type=int, default=512)
    ap.add_argument("--epochs", type=int, default=521)
    ap.add_argument("--lr", type=float, default=1e-3)
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--hip-graph", action="store_true")
    ap.add_argument("--prefetch", type=int, default=2, help="batches collated and copied to the GPU ahead (0 = inline)")
    ap.add_argument("--host-loader", action="store_true",
                    help="collate every batch on the host (as the reference does) instead of keeping the training "
                         "set in GPU memory and gathering batches there")
    ap.add_argument("--save-every", type=int, default=0)
    ap.add_argument("--text", action="store_true",
                    help="train the joint graph+text model (gtvae.py) instead of the graph-only step; synthetic "
                         "graphs then carry synthetic sentences, MRP input needs a 'tokens' id list per graph")
    ap.add_argument("--vocab", type=int, default=30522, help="vocabulary size of the text branch (bert-base-uncased)")
    ap.add_argument("--train-graph-recon", action="store_true",
                    help="joint model: also optimise the graph reconstruction loss (gtvae.py:124 restored)")
    args = ap.parse_args(argv)

    rank, local_rank, world = init_process_group()
    dev_ = torch.device("cuda", local_device_index(local_rank))
    torch.cuda.set_device(dev_)
    if args.train:
        trn = prep_data_list(load_mrp(args.train), args.nodes)
        val = prep_data_list(load_mrp(args.dev), args.nodes) if args.dev else []
    else:
        n = args.synthetic or 4096
        vocab = args.vocab if args.text else 0
        trn = synth_graph_list(n, args.nodes, seed=1234, vocab=vocab)
        val = synth_graph_list(max(n // 8, args.batch), args.nodes, seed=4321, vocab=vocab)
    if args.text:
        from .joint_trainer import JointTrainer

        if any(d.tokens is None for d in trn + val):
            raise SystemExit("--text needs token ids for every graph ('tokens' in the MRP lines)")
        trn_loader = DataLoader(trn, args.batch * world, args.nodes)
        val_loader = DataLoader(val, args.batch, args.nodes) if val else None
        jt = JointTrainer(args.nodes, args.latent, args.hid, args.nodes, args.vocab, 0, device=dev_, lr=args.lr,
                          seed=args.seed, train_graph_recon=args.train_graph_recon)
        gen = torch.Generator(device=dev_).manual_seed(args.seed + rank)
        return train_joint(trn_loader, val_loader, jt, args.epochs, rank=rank, world=world, generator=gen,
                           save_every=args.save_every)
    val_loader = DataLoader(val, args.batch, args.nodes) if val else None
    if not args.host_loader:
        from .data import DeviceDataset, DeviceLoader

        trn_loader = DeviceLoader(DeviceDataset(trn, args.nodes, dev_), args.batch * world, rank=rank, world=world)
    else:
        trn_loader = DataLoader(trn, args.batch * world, args.nodes)
        if world > 1:
            trn_loader = ShardedLoader(trn_loader, rank, world)
    if args.host_loader and args.prefetch > 0:
        from .data import DevicePrefetcher

        sharded = getattr(trn_loader, "sharded", False)
        trn_loader = DevicePrefetcher(trn_loader, dev_, args.prefetch)
        trn_loader.sharded = sharded
    trainer = GraphVAETrainer(args.nodes, args.latent, args.hid, args.nodes, device=dev_, lr=args.lr, seed=args.seed,
                              max_batch=args.batch, max_edges=args.batch * args.nodes * 6)
    gen = torch.Generator(device=dev_).manual_seed(args.seed + rank)
    return train(trn_loader, val_loader, trainer, args.epochs, rank=rank, world=world, generator=gen,
                 use_graph=args.hip_graph, save_every=args.save_every)


if __name__ == "__main__":
    main()
