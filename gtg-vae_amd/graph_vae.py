"""GraphVAE with the reference's class surface, running on HIP kernels.

Mirrors /root/reference/src/graph_vae.py: ``GraphVAE`` (:13-92),
``VGAE_encoder`` (:95-110) and ``MLP_VAE_plain`` (:113-131) keep their
constructor arguments, method names, return conventions and ``state_dict``
keys (``encoder.conv{1,2}.lin.weight``, ``encoder.conv{1,2}.bias``,
``decoder.decode_{1,2}.{weight,bias}``).  torch_geometric is not used: the
two GCNConv layers and global_max_pool are one fused encoder call
(functional.EncoderFn); the per-graph Python loss loop of graph_vae.py:42-55
is one batched kernel; the MST read-out is one wavefront per graph.
"""
import math
from typing import List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import _lib
from . import functional as Fn
from .data import gold_bitmask
from .maxspantree import mst_graph, mst_graph_batched

MAX_LOGSTD = 10
device = torch.device("cuda" if torch.cuda.is_available() else "cpu")


class GCNConv(nn.Module):
    """Parameter holder with torch_geometric.nn.GCNConv's names and default
    initialisation (glorot weight, zero bias).  The arithmetic lives in
    functional.EncoderFn, which consumes both layers at once."""

    def __init__(self, in_channels: int, out_channels: int):
        super().__init__()
        self.in_channels, self.out_channels = in_channels, out_channels
        self.lin = nn.Linear(in_channels, out_channels, bias=False)
        self.bias = nn.Parameter(torch.zeros(out_channels))
        self.reset_parameters()

    def reset_parameters(self):
        a = math.sqrt(6.0 / (self.in_channels + self.out_channels))
        with torch.no_grad():
            self.lin.weight.uniform_(-a, a)
            self.bias.zero_()

    def _load_from_state_dict(self, state_dict, prefix, *args, **kwargs):
        # torch_geometric 1.x stored `weight` as (in, out); 2.x stores `lin.weight` (out, in)
        old = prefix + "weight"
        if old in state_dict and prefix + "lin.weight" not in state_dict:
            state_dict[prefix + "lin.weight"] = state_dict.pop(old).t().contiguous()
        super()._load_from_state_dict(state_dict, prefix, *args, **kwargs)


class VGAE_encoder(nn.Module):
    def __init__(self, in_channels, out_channels):
        super().__init__()
        self.hidden_channels = out_channels * 2
        self.conv1 = GCNConv(in_channels, self.hidden_channels)
        self.relu = nn.ReLU()
        self.conv2 = GCNConv(self.hidden_channels, out_channels)

    def forward(self, x, edge_index, num_graphs: int = None, nmax: int = None, adj: torch.Tensor = None):
        """Pooled encoding (B, out_channels).  Unlike the reference's encoder
        (graph_vae.py:106-110) this returns the max-pooled rows: conv1, ReLU,
        conv2 and the pool are one fused call, the (B*N, out) node matrix is
        never written to memory.  ``x`` may be None for stacked identities."""
        if nmax is None or num_graphs is None:
            raise _lib.GtgError("VGAE_encoder.forward needs num_graphs and nmax (graphs are padded to nmax nodes)")
        extents = None
        if adj is None:
            adj, extents = Fn.build_adj(edge_index, num_graphs, nmax, return_extents=True)
        pooled, self.last_argmax = Fn.EncoderFn.apply(adj, x, self.conv1.lin.weight, self.conv1.bias,
                                                      self.conv2.lin.weight, self.conv2.bias, extents,
                                                      getattr(self, "matrix_mode", None))
        return pooled


class MLP_VAE_plain(nn.Module):
    def __init__(self, input_dim, hidden_dim, output_dim, nmax):
        super().__init__()
        self.decode_1 = nn.Linear(input_dim, hidden_dim)
        self.relu = nn.ReLU()
        self.decode_2 = nn.Linear(hidden_dim, output_dim)
        self.sigmoid = nn.Sigmoid()
        self.nmax = nmax
        for m in self.modules():
            if isinstance(m, nn.Linear):
                nn.init.xavier_uniform_(m.weight, gain=nn.init.calculate_gain("relu"))

    def forward(self, h):
        return Fn.DecoderFn.apply(h, self.decode_1.weight, self.decode_1.bias, self.decode_2.weight, self.decode_2.bias)


class GraphVAE(nn.Module):
    def __init__(self, input_dim, latent_dim, mlp_hid_dim, nmax, phase):
        super().__init__()
        if nmax > 64:
            raise _lib.GtgError("nmax > 64 is not supported by the one-wavefront-per-graph kernels")
        self.phase = phase
        self.nmax = nmax
        self.input_dim = input_dim
        output_dim = nmax * (nmax + 1) // 2
        self.encoder = VGAE_encoder(input_dim, out_channels=latent_dim)
        self.decoder = MLP_VAE_plain(latent_dim, mlp_hid_dim, output_dim, nmax)

    # -- graph_vae.py:22-28 --------------------------------------------------
    def encode(self, batch, x, edge_index, adj: torch.Tensor = None):
        """``batch`` is the (B*nmax,) graph-id vector of the collated batch (or
        the number of graphs as an int); every graph is padded to nmax rows,
        so B = len(batch) / nmax and no device read-back is needed.  ``x`` is
        the (B*nmax, input_dim) feature matrix, or None for the stacked
        identities the training loop always feeds (train.py:203-204)."""
        num_graphs = int(batch) if isinstance(batch, int) else batch.numel() // self.nmax
        self.__h__ = self.encoder(x, edge_index, num_graphs=num_graphs, nmax=self.nmax, adj=adj)
        return self.__h__

    def decode(self, z):
        return self.decoder(z)

    # -- graph_vae.py:34-59 --------------------------------------------------
    def loss(self, out, adj, gold_edges, report):
        """Returns (graph_recon_loss, edge_recall, edge_precision) like the
        reference: the loss carries the autograd graph; recall/precision are
        0-dim CPU tensors (zero when ``report`` is false).  ``adj`` may be the
        reference's list of (N,N) numpy arrays or a (B,N,N) tensor;
        ``gold_edges`` a list of sets or a (B,N) int64 bit-mask tensor."""
        if not torch.is_tensor(adj):
            adj = torch.from_numpy(np.asarray(adj))
        adj = adj.to(device=out.device, dtype=torch.float32)
        B = adj.shape[0]
        graph_recon_loss = Fn.BCELossFn.apply(out, adj)
        total_edge_recall = torch.tensor(0).float()
        total_edge_precision = torch.tensor(0).float()
        if report:
            rec, prec = self.batched_statistics(out.detach(), gold_edges)
            # the reference accumulates per-graph values sequentially in fp32 (graph_vae.py:54-55)
            r32, p32 = np.float32(0), np.float32(0)
            for r, p in zip(rec.cpu().numpy(), prec.cpu().numpy()):
                r32 = np.float32(r32 + r)
                p32 = np.float32(p32 + p)
            total_edge_recall = torch.tensor(r32)
            total_edge_precision = torch.tensor(p32)
        total_edge_recall = total_edge_recall / B
        total_edge_precision = total_edge_precision / B
        return graph_recon_loss, total_edge_recall, total_edge_precision

    def batched_statistics(self, out, gold_edges):
        """Per-graph (recall, precision) on the GPU: batched MST on the decoder
        rows + bit-mask intersection (graph_vae.py:75-92 for every graph at once)."""
        if torch.is_tensor(gold_edges):
            bits = gold_edges.to(out.device)
        else:
            bits = gold_bitmask([set(g) for g in gold_edges], self.nmax).to(out.device)
        edges, count = mst_graph_batched(out.contiguous(), self.nmax)
        return Fn.edge_precision_recall(edges, count, bits, self.nmax)

    # -- graph_vae.py:61-69 --------------------------------------------------
    def recover_adj_lower(self, l):
        adj = torch.zeros(self.nmax, self.nmax, device=l.device, dtype=l.dtype)
        iu = torch.triu_indices(self.nmax, self.nmax, device=l.device)
        adj[iu[0], iu[1]] = l.reshape(-1)
        return adj

    def recover_full_adj_from_lower(self, lower):
        diag = torch.diag(torch.diag(lower, 0))
        return lower + torch.transpose(lower, 0, 1) - diag

    # -- graph_vae.py:71-73 --------------------------------------------------
    def adj_recon_loss(self, adj_pred, adj_truth):
        """BCE between one predicted upper-triangular vector and its target vector."""
        target = self.recover_adj_lower(adj_truth.to(adj_pred.device).float())
        return Fn.BCELossFn.apply(adj_pred.reshape(1, -1), target.unsqueeze(0))

    # -- graph_vae.py:75-92 --------------------------------------------------
    def graph_statistics(self, adj_matrix, gold_edges, report):
        pred_edges = mst_graph(adj_matrix)
        gold_list = list(gold_edges) if gold_edges is not None else []
        gold = set(tuple(e) for e in gold_list)
        count = sum(1 for j in pred_edges if (j[0], j[1]) in gold or (j[1], j[0]) in gold)
        if len(gold_list) == 0 or len(pred_edges) == 0:
            return torch.tensor(1), torch.tensor(1)
        return count / len(gold_list), count / len(pred_edges)
