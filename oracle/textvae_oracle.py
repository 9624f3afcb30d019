"""CPU oracle for the text branch and the joint graph+text step.  TEST INFRASTRUCTURE ONLY.

Only tests/ and tools that time a CPU baseline may import this module; the
product path (gtg_vae_amd) never does.

Functional eager-PyTorch-CPU restatement, over a plain dict of tensors keyed as
the reference's ``state_dict``, of /root/reference/src/text_vae.py in
evaluation mode (dropout and word noise off, text_vae.py:55-57, 62-64, 69-72):

* ``lstm``        <- nn.LSTM(512, 1024, 1) as used at text_vae.py:37-40, written
                     out as the cell recurrence (gate order i, f, g, o;
                     c' = f*c + i*g, h' = o*tanh(c')), so that the product's
                     fused recurrences are checked against explicit arithmetic
                     and not against another library call
* ``encode``      <- DAE.encode (text_vae.py:54-59): embed, bidirectional LSTM,
                     hgen on [h_fwd_last ; h_bwd_last]
* ``decode``      <- DAE.decode (text_vae.py:61-66): embed + z2emb(z), LSTM, proj
* ``loss_rec``    <- DAE.loss_rec (text_vae.py:77-79)
* ``accuracy``    <- DAE.accuracy (text_vae.py:81-101)
* ``joint_loss``  <- GTVAE.forward (/root/reference/src/gtvae.py:95-126) composed
                     from these and the graph oracle's pieces: joint_h = (text_h +
                     graph_h)/2, loss = kl_w*KL + text_recon (+ graph_recon when
                     the line the reference leaves commented out, gtvae.py:124, is
                     restored by ``train_graph_recon``)
* ``forward_1``   <- GTVAE.forward_1 (gtvae.py:77-91)
* ``word_drop``   <- noise.word_drop (/root/reference/src/noise.py:12-22) with the
                     keep flags it draws per sentence (noise.py:16) passed in
* ``embed_dropout`` <- self.drop(self.embed(input)) [+ self.z2emb(z)]
                     (text_vae.py:56, 62) with the dropout mask passed in

Pinned: tests/test_oracle_textvae.py checks encode / decode / loss_rec /
accuracy against tests/golden/textvae_golden.npz, which was produced by running
the reference's own text_vae.DAE (tests/golden/make_text_golden.py), and
word_drop against tests/golden/noise_golden.npz, produced by running the
reference's own noise.word_drop (tests/golden/make_noise_golden.py).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
import torch.nn.functional as F

from . import graphvae_oracle as G


def lstm(x, w_ih, w_hh, b_ih, b_hh, h0=None, c0=None, reverse: bool = False):
    """x (T, B, in) -> (outputs (T, B, H), h_T, c_T); one layer, one direction."""
    T, B, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H) if h0 is None else h0
    c = x.new_zeros(B, H) if c0 is None else c0
    xw = F.linear(x, w_ih, b_ih)  # the input half of the gates for all steps at once
    outs = [None] * T
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        gates = xw[t] + F.linear(h, w_hh, b_hh)
        i, f, g, o = gates.chunk(4, dim=1)
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(g)
        h = torch.sigmoid(o) * torch.tanh(c)
        outs[t] = h
    return torch.stack(outs), h, c


def _p(params: Dict[str, torch.Tensor], prefix: str, name: str) -> torch.Tensor:
    return params[prefix + name]


def encode(params, inputs, prefix: str = ""):
    """inputs (T, B) int64 -> (B, latent)."""
    emb = F.embedding(inputs, _p(params, prefix, "embed.weight"))
    _, hf, _ = lstm(emb, *(_p(params, prefix, "E.%s_l0" % k) for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")))
    _, hb, _ = lstm(emb, *(_p(params, prefix, "E.%s_l0_reverse" % k)
                           for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")), reverse=True)
    return F.linear(torch.cat([hf, hb], 1), _p(params, prefix, "hgen.weight"), _p(params, prefix, "hgen.bias"))


def decode(params, z, inputs, prefix: str = "", hidden=None):
    """-> (logits (T, B, V), (h_T, c_T))."""
    emb = F.embedding(inputs, _p(params, prefix, "embed.weight")) + \
        F.linear(z, _p(params, prefix, "z2emb.weight"), _p(params, prefix, "z2emb.bias"))
    h0, c0 = (None, None) if hidden is None else hidden
    out, h, c = lstm(emb, *(_p(params, prefix, "G.%s_l0" % k) for k in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")),
                     h0=h0, c0=c0)
    logits = F.linear(out.reshape(-1, out.shape[-1]), _p(params, prefix, "proj.weight"), _p(params, prefix, "proj.bias"))
    return logits.view(out.shape[0], out.shape[1], -1), (h, c)


def loss_rec(logits, targets, padidx: int):
    loss = F.cross_entropy(logits.reshape(-1, logits.shape[-1]), targets.reshape(-1), ignore_index=padidx,
                           reduction="none").view(targets.shape)
    return loss.sum(dim=0)


def accuracy(logits, targets, padidx: int) -> float:
    pred = torch.argmax(torch.softmax(logits, 2), 2)
    mask = targets != padidx
    return float(((pred == targets) & mask).sum() / mask.sum())


def shift_targets(dec_seq):
    """(B, T) token ids as train.py feeds them -> (inputs (T, B), targets (T, B)) (gtvae.py:101-104)."""
    B = dec_seq.shape[0]
    pad = torch.zeros(B, 1, dtype=dec_seq.dtype)
    return dec_seq.t(), torch.cat((dec_seq[:, 1:], pad), 1).long().t()


def joint_loss(params, edge_index, adj, dec_seq, eps, kl_w: float, num_graphs: int, nmax: int, padidx: int = 0,
               train_graph_recon: bool = False):
    """params: a GTVAE state dict (keys graph_vae.*, text_vae.*, zg_*).  Returns a dict with the optimised
    ``loss`` and the reported pieces."""
    gp = {k[len("graph_vae."):] if k.startswith("graph_vae.") else k: v for k, v in params.items()
          if not k.startswith("text_vae.")}
    inputs, targets = shift_targets(dec_seq)
    graph_h = G.encode(gp, None, edge_index, num_graphs, nmax)
    text_h = encode(params, inputs, "text_vae.")
    joint_h = (text_h + graph_h) / 2
    z, mu, logvar, kl = G.latent(gp, joint_h, eps)
    logits, _ = decode(params, z, inputs, "text_vae.")
    text_rec = loss_rec(logits, targets, padidx).mean()
    out = G.decode(gp, z)
    graph_rec = G.recon_loss(out, adj)
    loss = kl_w * kl + text_rec
    if train_graph_recon:
        loss = loss + graph_rec
    return {"loss": loss, "kl": kl, "text_recon": text_rec, "graph_recon": graph_rec,
            "text_acc": accuracy(logits.detach(), targets, padidx), "out": out}


def forward_1(params, inputs, targets, eps, padidx: int = 0):
    """GTVAE.forward_1: the text-only VAE step through the shared zg_mu / zg_var head."""
    text_h = encode(params, inputs, "text_vae.")
    z, mu, logvar, kl = G.latent(params, text_h, eps)
    logits, _ = decode(params, z, inputs, "text_vae.")
    rec = loss_rec(logits, targets, padidx).mean()
    return kl + rec, accuracy(logits.detach(), targets, padidx)


def word_drop(x, keep, padidx: int):
    """noise.word_drop (noise.py:12-22) for token ids x (T, B) with the keep flags of noise.py:16 given as
    ``keep`` (T, B) bool: per sentence the first token is always kept (noise.py:17), kept tokens stay in order
    (noise.py:18) and the sentence is padded back to its length with padidx (noise.py:19)."""
    x = torch.as_tensor(x)
    keep = torch.as_tensor(keep).bool()
    cols = []
    for i in range(x.size(1)):
        words = x[:, i].tolist()
        k = keep[:, i].tolist()
        k[0] = True
        sent = [w for j, w in enumerate(words) if k[j]]
        sent += [padidx] * (len(words) - len(sent))
        cols.append(sent)
    return torch.tensor(cols, dtype=torch.long).t().contiguous()


def embed_dropout(table, ids, keep=None, p: float = 0.0, add=None):
    """drop(table[ids]) (+ add broadcast over time), text_vae.py:56 / :62, with dropout's keep mask (T, B, D)
    given: kept values are scaled by 1 / (1 - p) as nn.Dropout does in training mode."""
    out = table[ids]
    if keep is not None:
        out = out * keep.to(out.dtype) / (1.0 - p)
    if add is not None:
        out = out + add.unsqueeze(0)
    return out
