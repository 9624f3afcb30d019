"""Joint graph+text VAE with the reference's class surface.

Mirrors /root/reference/src/gtvae.py:15-126.  The graph half (encode, latent
head, reparameterisation, KL, decode, reconstruction loss) runs on the HIP
kernels; the text half is text_vae.DAE: stock PyTorch-ROCm embeddings and LSTMs,
HIP vocabulary projection + cross-entropy (DAE.decode_loss).

Two deliberate, flag-selectable differences from the file as shipped:
* ``text=False`` builds no text branch; joint_h is then the graph encoding
  alone (the reference always averages both encodings, gtvae.py:107).
* ``train_graph_recon`` adds the graph reconstruction loss to the optimised
  loss; the reference computes it but leaves ``loss += graph_recon_loss``
  commented out (gtvae.py:124).  Default False = the reference's behaviour
  when ``text=True``; graph-only models default to True (otherwise nothing but
  the KL term would train).
"""
import torch
import torch.nn as nn

from . import functional as Fn
from .graph_vae import GraphVAE
from .text_vae import DAE


class GTVAE(nn.Module):
    def __init__(self, input_dim, latent_dim, mlp_hid_dim, max_num_nodes, vocabsize=None, padidx=0, phase=1,
                 text=True, train_graph_recon=None):
        super().__init__()
        self.graph_vae = GraphVAE(input_dim, latent_dim, mlp_hid_dim, max_num_nodes, 1)
        self.text_vae = DAE(latent_dim, vocabsize, padidx) if text else None
        self.zg_mu = nn.Linear(latent_dim, latent_dim)
        self.zg_var = nn.Linear(latent_dim, latent_dim)
        self.max_num_nodes = max_num_nodes
        self.phase = phase
        self.train_graph_recon = (not text) if train_graph_recon is None else train_graph_recon
        self.eps_generator = None  # optional torch.Generator for reproducible latent noise
        # True: text accuracy is returned as a Python float like the reference's (a host sync per call);
        # False: as a 0-dim device tensor
        self.host_metrics = True

    def _eps(self, like):
        return torch.randn(like.shape, dtype=like.dtype, device=like.device, generator=self.eps_generator)

    # gtvae.py:33-39 — samples in training AND evaluation mode (`if True:`)
    def reparameterize(self, mu, logvar, eps=None):
        z, _ = Fn.ReparamKLFn.apply(mu, logvar, self._eps(mu) if eps is None else eps)
        return z

    # gtvae.py:41-42
    def loss_kl(self, mu, logvar):
        _, kl = Fn.ReparamKLFn.apply(mu, logvar, torch.zeros_like(mu))
        return kl

    # gtvae.py:77-91
    def forward_1(self, inputs, targets, eps=None):
        text_h = self.text_vae.autoenc(inputs, targets)
        text_z, text_kl_loss, _, _ = Fn.LatentFn.apply(text_h, self.zg_mu.weight, self.zg_mu.bias, self.zg_var.weight,
                                                       self.zg_var.bias, self._eps(text_h) if eps is None else eps)
        tok_loss, pred, _ = self.text_vae.decode_loss(text_z, inputs, targets)
        text_recon_loss = tok_loss.sum(dim=0).mean()
        text_acc = self.text_vae.accuracy_from_pred(pred, targets, self.host_metrics)
        return text_kl_loss + text_recon_loss, text_acc

    # gtvae.py:95-126
    def forward(self, x, edge_index, batch, adj, gold_edges, dec_seq, kl_anneal_w, is_dev=False, eps=None,
                report=False):
        text_acc, _text_acc, _edge_recall, _edge_precision = 0, 0, 0, 0
        text_recon_loss = 0
        graph_h = self.graph_vae.encode(batch, x, edge_index)
        if self.text_vae is not None:
            B, _ = dec_seq.size()
            pad_tokens = torch.full((B, 1), 0, device=dec_seq.device, dtype=dec_seq.dtype)
            inputs = dec_seq.t()
            targets = torch.cat((dec_seq[:, 1:], pad_tokens), 1).long().t()
            text_h = self.text_vae.autoenc(inputs, targets)
            joint_h = (text_h + graph_h) / 2
        else:
            joint_h = graph_h
        joint_z, joint_kl_loss, self.joint_mu, self.joint_logvar = Fn.LatentFn.apply(
            joint_h, self.zg_mu.weight, self.zg_mu.bias, self.zg_var.weight, self.zg_var.bias,
            self._eps(joint_h) if eps is None else eps)
        loss = joint_kl_loss * kl_anneal_w
        if self.text_vae is not None:
            tok_loss, pred, _ = self.text_vae.decode_loss(joint_z, inputs, targets)
            text_recon_loss = tok_loss.sum(dim=0).mean()
            text_acc = self.text_vae.accuracy_from_pred(pred, targets, self.host_metrics)
            loss = loss + text_recon_loss
        out_tensor = self.graph_vae.decode(joint_z)
        graph_recon_loss, edge_recall, edge_precision = self.graph_vae.loss(out_tensor, adj, gold_edges, report)
        if self.train_graph_recon:
            loss = loss + graph_recon_loss
        return (loss, text_acc, joint_kl_loss, graph_recon_loss, text_recon_loss, edge_recall, edge_precision,
                _text_acc, _edge_recall, _edge_precision)
