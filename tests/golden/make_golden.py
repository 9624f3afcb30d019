"""Generate the golden fixtures under tests/golden/ by RUNNING the reference.

Run in the build container only (needs /root/reference):

    python tests/golden/make_golden.py

Outputs (committed):
  mst_golden.npz       inputs (N,N) float32 / triu rows and the ordered edge
                       lists returned by the reference's maxspantree.mst_graph
  graphvae_golden.npz  MLP_VAE_plain weights, inputs and outputs; triu
                       vectorise / recover helpers; binary_cross_entropy values;
                       graph_statistics precision/recall

maxspantree.py imports with torch alone.  graph_vae.py imports torch_geometric
and sklearn at module level; torch_geometric is not installed, so placeholder
module objects are registered in sys.modules purely so that the import
statement succeeds.  Nothing produced here touches GCNConv / global_max_pool:
only MLP_VAE_plain and the pure-torch helper methods are executed, so the
placeholders contribute no arithmetic to any fixture.
"""
import os
import sys
import types

import numpy as np
import torch

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def _import_reference():
    sys.path.insert(0, REF)
    import maxspantree  # noqa

    tg = types.ModuleType("torch_geometric")
    tgu = types.ModuleType("torch_geometric.utils")
    tgn = types.ModuleType("torch_geometric.nn")
    for name in ("negative_sampling", "remove_self_loops", "add_self_loops"):
        setattr(tgu, name, None)

    class _Placeholder(torch.nn.Module):  # never called
        def __init__(self, *a, **k):
            super().__init__()

    tgn.GCNConv = _Placeholder
    tgn.global_max_pool = None
    tg.utils, tg.nn = tgu, tgn
    sys.modules.setdefault("torch_geometric", tg)
    sys.modules.setdefault("torch_geometric.utils", tgu)
    sys.modules.setdefault("torch_geometric.nn", tgn)
    import graph_vae  # noqa

    return maxspantree, graph_vae


def sym(rng, n, lo=0.0, hi=1.0):
    a = rng.uniform(lo, hi, size=(n, n)).astype(np.float32)
    a = np.triu(a) + np.triu(a, 1).T
    return a.astype(np.float32)


def mst_cases(rng):
    cases = []

    def add(name, G):
        cases.append((name, np.ascontiguousarray(G, dtype=np.float32)))

    add("doc_sample", [[0, 2, 0, 6, 0], [2, 0, 3, 8, 5], [0, 3, 0.6, 0, 7], [6, 8, 0, 0, 9], [0, 5, 7, 9, 0.7]])
    add("clique3_half", np.full((3, 3), 0.5) + np.eye(3) * 0.5)
    add("diag_exactly_half", np.full((4, 4), 0.7) - np.eye(4) * 0.2)
    add("identity3", np.eye(3))
    add("blockdisc4", [[1, 0, 0, 0], [0, 1, 0, 5], [0, 0, 1, 0], [0, 5, 0, 1]])
    add("neg_huge", [[1, -1e19, 0.3], [-1e19, 1, 0.2], [0.3, 0.2, 1]])
    add("nan_w", [[1, np.nan, 0.3], [np.nan, 1, 0.2], [0.3, 0.2, 1]])
    add("inf_w", [[1, -np.inf, 0.3], [-np.inf, 1, np.inf], [0.3, np.inf, 1]])
    add("neg_weights", [[1, -0.5, -0.25, -3], [-0.5, 1, -0.125, -2], [-0.25, -0.125, 1, -1], [-3, -2, -1, 1]])
    add("neg_zero", [[1, -0.0, 0.3], [-0.0, 1, -0.0], [0.3, -0.0, 1]])
    add("nan_diag", [[np.nan, 0.9, 0.3], [0.9, 1, 0.2], [0.3, 0.2, 1]])
    for n in (1, 2, 3, 10, 32, 64):
        for rep in range(3):
            G = sym(rng, n)
            np.fill_diagonal(G, rng.uniform(0.51, 1.0, size=n))
            add(f"allpresent_n{n}_{rep}", G)
        for rep in range(3):
            G = sym(rng, n)
            d = rng.uniform(0.3, 1.0, size=n).astype(np.float32)
            if n > 2:
                d[rng.integers(0, n)] = 0.5
            np.fill_diagonal(G, d)
            add(f"partial_n{n}_{rep}", G)
        G = np.round(sym(rng, n) * 8) / 8  # heavy ties, exact zeros
        np.fill_diagonal(G, rng.choice([0.25, 0.75, 1.0], size=n))
        add(f"ties_n{n}", G)
        G = sym(rng, n)
        np.fill_diagonal(G, 0.1)
        add(f"nonepresent_n{n}", G)
        G = rng.uniform(0, 1, size=(n, n)).astype(np.float32)  # NOT symmetric
        G[rng.uniform(size=(n, n)) < 0.3] = 0
        np.fill_diagonal(G, rng.uniform(0.4, 1.0, size=n))
        add(f"nonsym_n{n}", G)
        if n >= 4:
            # several components: exercises the "-1 means last vertex" path
            G = sym(rng, n)
            comp = rng.integers(0, 3, size=n)
            G[comp[:, None] != comp[None, :]] = 0
            np.fill_diagonal(G, rng.uniform(0.45, 1.0, size=n))
            add(f"components_n{n}", G)
            G2 = G.copy()
            G2[:, n - 1] = 0
            G2[n - 1, :] = 0
            G2[n - 1, n - 1] = 1.0
            add(f"isolated_last_n{n}", G2)
    return cases


def main():
    mst, gv = _import_reference()
    rng = np.random.default_rng(20240607)
    out = {}
    names = []
    for name, G in mst_cases(rng):
        edges = mst.mst_graph(torch.from_numpy(G.copy()))
        e = np.asarray(edges, dtype=np.int32).reshape(-1, 2)
        out[f"G__{name}"] = G
        out[f"E__{name}"] = e
        names.append(name)

    # triu-row path: decoder row -> recover_adj_lower -> recover_full -> mst_graph
    tri_names = []
    for n in (3, 10, 32, 64):
        model = gv.GraphVAE(n, 8, 16, n, 1)
        M = n * (n + 1) // 2
        for rep in range(3):
            row = torch.from_numpy(rng.uniform(0, 1, size=(1, M)).astype(np.float32))
            if rep == 2:
                row = torch.round(row * 4) / 4
            full = model.recover_full_adj_from_lower(model.recover_adj_lower(row))
            edges = mst.mst_graph(full)
            nm = f"triu_n{n}_{rep}"
            out[f"R__{nm}"] = row.numpy()[0]
            out[f"F__{nm}"] = full.numpy()
            out[f"E__{nm}"] = np.asarray(edges, dtype=np.int32).reshape(-1, 2)
            tri_names.append(nm)
    # the same path with diagonal entries that the expansion U + U^T - diag(U) (graph_vae.py:67-69) does not
    # return unchanged: +inf becomes inf + inf - inf = NaN (vertex absent), a value above FLT_MAX / 2 becomes
    # inf - d = inf (present), NaN stays NaN; plus infinite and NaN weights off the diagonal.  A generator of
    # its own, so that the cases above keep their values.
    rng2 = np.random.default_rng(20261111)
    for n in (3, 10, 64):
        model = gv.GraphVAE(n, 8, 16, n, 1)
        M = n * (n + 1) // 2
        diag_pos = np.array([i * n - (i * (i - 1)) // 2 for i in range(n)])
        for rep, special in enumerate(([np.inf], [3.0e38], [np.inf, 3.0e38, np.nan, -np.inf, 0.5, 0.75])):
            row = rng2.uniform(0, 1, size=(1, M)).astype(np.float32)
            hit = rng2.choice(n, size=max(1, n // 3), replace=False)
            row[0, diag_pos[hit]] = rng2.choice(np.array(special, dtype=np.float32), size=len(hit))
            if rep == 2:
                off = np.setdiff1d(np.arange(M), diag_pos)
                pick = rng2.choice(off, size=max(1, M // 10), replace=False)
                row[0, pick] = rng2.choice(np.array([np.inf, -np.inf, np.nan, 0.0], dtype=np.float32), size=len(pick))
            row = torch.from_numpy(row)
            full = model.recover_full_adj_from_lower(model.recover_adj_lower(row))
            edges = mst.mst_graph(full)
            nm = f"triu_diag_n{n}_{rep}"
            out[f"R__{nm}"] = row.numpy()[0]
            out[f"F__{nm}"] = full.numpy()
            out[f"E__{nm}"] = np.asarray(edges, dtype=np.int32).reshape(-1, 2)
            tri_names.append(nm)
    out["names"] = np.array(names)
    out["triu_names"] = np.array(tri_names)
    np.savez_compressed(os.path.join(HERE, "mst_golden.npz"), **out)
    print("mst cases:", len(names), "triu cases:", len(tri_names))

    # ---------------- graph_vae helpers ----------------
    gout = {}
    torch.manual_seed(7)
    for tag, (n, d, h, B) in {"a": (10, 32, 512, 6), "b": (7, 8, 24, 5)}.items():
        model = gv.GraphVAE(n, d, h, n, 1)
        dec = model.decoder
        for k, v in dec.state_dict().items():
            gout[f"{tag}__dec__{k}"] = v.numpy().copy()
        z = torch.randn(B, d)
        with torch.no_grad():
            p = model.decode(z)
        gout[f"{tag}__z"] = z.numpy()
        gout[f"{tag}__out"] = p.numpy()
        # targets as GraphVAE.loss builds them (graph_vae.py:48) + per-graph BCE (graph_vae.py:50,73)
        adj = (torch.rand(B, n, n) < 0.3).float()
        adj = ((adj + adj.transpose(1, 2)) > 0).float()
        vec = torch.stack([adj[b][torch.triu(torch.ones(n, n)) == 1] for b in range(B)])
        losses = torch.stack([model.adj_recon_loss(p[b], vec[b]) for b in range(B)])
        gout[f"{tag}__adj"] = adj.numpy()
        gout[f"{tag}__vec"] = vec.numpy()
        gout[f"{tag}__bce_per_graph"] = losses.numpy()
        gout[f"{tag}__bce_mean"] = (losses.sum() / B).numpy()
        # gradient of the mean loss w.r.t. z through decoder (autograd of the reference modules)
        z2 = z.clone().requires_grad_(True)
        p2 = model.decode(z2)
        loss = sum(model.adj_recon_loss(p2[b], vec[b]) for b in range(B)) / B
        loss.backward()
        gout[f"{tag}__dz"] = z2.grad.numpy()
        for k, prm in dec.named_parameters():
            gout[f"{tag}__grad__{k}"] = prm.grad.numpy().copy()
        # recover helpers
        full = torch.stack([model.recover_full_adj_from_lower(model.recover_adj_lower(p[b:b + 1])) for b in range(B)])
        gout[f"{tag}__full"] = full.numpy()
    # BCE clamp behaviour at the ends of the range
    pe = torch.tensor([0.0, 1.0, 1e-45, 1 - 6e-8, 0.5, 0.0, 1.0])
    te = torch.tensor([1.0, 0.0, 1.0, 0.0, 1.0, 0.0, 1.0])
    gout["edge__p"] = pe.numpy()
    gout["edge__t"] = te.numpy()
    gout["edge__bce_each"] = torch.nn.functional.binary_cross_entropy(pe, te, reduction="none").numpy()
    gout["edge__bce"] = gv.GraphVAE.adj_recon_loss(None, pe, te).numpy()

    # graph_statistics (graph_vae.py:75-92)
    model = gv.GraphVAE(10, 8, 16, 10, 1)
    stats_in, stats_gold, stats_out = [], [], []
    for rep in range(6):
        G = sym(rng, 10)
        np.fill_diagonal(G, rng.uniform(0.3, 1.0, size=10))
        ne = int(rng.integers(0, 9))
        gold = set()
        for _ in range(ne):
            a, b = rng.integers(0, 10, size=2)
            if a != b:
                gold.add((int(a), int(b)))
        if rep == 5:
            gold = set()
        r, p_ = model.graph_statistics(torch.from_numpy(G), list(gold), True)
        stats_in.append(G)
        gmat = np.zeros((10, 10), dtype=np.uint8)
        for a, b in gold:
            gmat[a, b] = 1
        stats_gold.append(gmat)
        stats_out.append([float(r), float(p_)])
    gout["stats__G"] = np.stack(stats_in)
    gout["stats__gold"] = np.stack(stats_gold)
    gout["stats__recall_precision"] = np.asarray(stats_out, dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "graphvae_golden.npz"), **gout)
    print("graphvae golden keys:", len(gout))


if __name__ == "__main__":
    main()
