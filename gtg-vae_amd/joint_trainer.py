"""Training step of the joint graph+text VAE (gtvae.py:95-126 driven as train.py:302-314 drives it) with
the optimiser and the data-parallel exchange of the graph-only fast path.

The forward and backward run through the GTVAE module (HIP kernels for the graph half, the LSTM
recurrences, the embedding lookup / dropout / word noise and the vocabulary projection + cross-entropy;
autograd connects them; the small Linear layers hgen and z2emb run on the library's GEMM as well).  What this class adds around the module:

* every parameter and every gradient lives in ONE flat fp32 buffer each (the module's parameters are views
  into it), so the optimiser is a single ``gtg_adam_step`` launch over all 65 M+ parameters (train.py:290:
  Adam(lr=1e-3, betas=(0.5, 0.999))) and the gradient exchange is a few large all-reduces instead of one
  per tensor;
* the flat buffer is ordered by when backward finishes each group, and cut into three buckets:
  the vocabulary projection (half of all parameters, final first) | decoder LSTM, z2emb, graph decoder,
  latent head | encoder LSTMs, hgen, embedding, graph encoder.  A bucket's sum all-reduce (RCCL) is
  launched from an autograd hook the moment its last gradient has been accumulated, so the projection's
  125 MB travel while the LSTM backward still runs;
* shards of unequal size are weighted by B_local * world / B_global and an empty shard contributes zeros
  (same rule as GraphVAETrainer).
"""
from typing import Dict, List, Optional

import torch

from . import _lib
from ._lib import check, ptr
from .data import GraphBatch
from .gtvae import GTVAE
from .parallel import GradBuckets, shard_weight

# parameter-name prefixes per bucket, in the order backward completes them
_BUCKET_PREFIXES = (
    ("text_vae.proj.",),
    ("text_vae.G.", "text_vae.z2emb.", "graph_vae.decoder.", "zg_mu.", "zg_var."),
    ("text_vae.hgen.", "text_vae.E.", "text_vae.embed.", "graph_vae.encoder."),
)


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


class JointTrainer:
    def __init__(self, input_dim: int, latent_dim: int, mlp_hid_dim: int, nmax: int, vocabsize: int, padidx: int = 0,
                 device="cuda", init: Optional[Dict[str, torch.Tensor]] = None, lr: float = 1e-3, betas=(0.5, 0.999),
                 eps: float = 1e-8, group=None, seed: int = 0, matrix_mode=None, train_graph_recon: bool = False):
        self.L = _lib.lib()
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise _lib.GtgError("JointTrainer needs a GPU device; there is no CPU path")
        self.n, self.d, self.hid, self.vocab, self.padidx = nmax, latent_dim, mlp_hid_dim, vocabsize, padidx
        self.lr, self.betas, self.eps = lr, betas, eps
        torch.manual_seed(seed)
        model = GTVAE(input_dim, latent_dim, mlp_hid_dim, nmax, vocabsize, padidx, text=True,
                      train_graph_recon=train_graph_recon)
        if init is not None:
            model.load_state_dict(init)
        mode = _lib.matrix_mode(matrix_mode)
        model.graph_vae.encoder.matrix_mode = mode
        model.text_vae.matrix_mode = mode
        model.host_metrics = False          # text accuracy stays a device tensor: no host sync per step
        self.model = model.to(self.dev)
        if not train_graph_recon:
            # as the reference ships (gtvae.py:124 commented out) the graph decoder receives no gradient
            for p in self.model.graph_vae.decoder.parameters():
                p.requires_grad_(False)

        # ---- flat parameter / gradient / moment buffers, bucket by bucket -----------------------------
        named = dict(self.model.named_parameters())
        order: List[List[str]] = []
        for prefixes in _BUCKET_PREFIXES:
            order.append([k for pre in prefixes for k in named if k.startswith(pre)])
        placed = {k for grp in order for k in grp}
        left = [k for k in named if k not in placed]
        if left:
            raise _lib.GtgError("parameters outside every gradient bucket: %s" % left)
        self.offsets: Dict[str, tuple] = {}
        bounds = [0]
        off = 0
        for grp in order:
            for k in grp:
                n = named[k].numel()
                self.offsets[k] = (off, n, tuple(named[k].shape))
                off = _align(off + n)
            bounds.append(off)
        self.numel = off
        self.flat_p = torch.zeros(self.numel, dtype=torch.float32, device=self.dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.adam_state = torch.zeros(4, dtype=torch.int32, device=self.dev)
        self.p: Dict[str, torch.Tensor] = {}
        self.g: Dict[str, torch.Tensor] = {}
        with torch.no_grad():
            for k, (o, n, shape) in self.offsets.items():
                view = self.flat_p[o:o + n].view(shape)
                view.copy_(named[k].data)
                named[k].data = view                      # the module now reads and Adam writes the same memory
                self.p[k] = view
                self.g[k] = self.flat_g[o:o + n].view(shape)
                named[k].grad = self.g[k]                 # autograd accumulates in place into the flat buffer
        # the embedding's gradient is scatter-added straight into its rows of the flat buffer (functional.
        # EmbedDropoutFn); autograd receives None for it, and still runs the parameter's post-accumulate hook
        # below once every use of the table has been differentiated, i.e. when the scatter-adds have been enqueued
        named["text_vae.embed.weight"].grad_inplace = True
        # the large weight gradients -- vocabulary projection, the LSTMs' W_ih / W_hh: 49 M of the text branch's
        # 51 M weights -- are stored by their backward straight into their views of the flat buffer
        # (functional._grad_slot): each is used by one call per pass and the buffer is cleared every step, so
        # storing equals accumulating, without autograd's `grad += new` pass over each of them (their
        # post-accumulate hooks fire all the same, after the backward that stored them has returned)
        self._stored = [k for k in named if k == "text_vae.proj.weight" or
                        (k.startswith(("text_vae.E.weight_", "text_vae.G.weight_")))]
        for k in self._stored:
            named[k].grad_overwrite = True
        self.buckets = GradBuckets(self.flat_g, bounds, group)
        self.world = self.buckets.world

        # ---- bucket launch from autograd: count down the gradients each bucket still waits for ---------
        self._bucket_of = {}
        self._expected = [0] * len(order)
        for b, grp in enumerate(order):
            for k in grp:
                if named[k].requires_grad:
                    self._bucket_of[k] = b
                    self._expected[b] += 1
                    named[k].register_post_accumulate_grad_hook(self._make_hook(b))
        self._waiting = list(self._expected)
        self._next_bucket = 0
        self._reduce_now = False

    # ------------------------------------------------------------------ hooks
    def _make_hook(self, b: int):
        def hook(_param):
            self._waiting[b] -= 1
            if self._reduce_now:
                self._launch_ready_buckets()
        return hook

    def _launch_ready_buckets(self, force: bool = False):
        """Start the all-reduce of every bucket that is final, strictly in index order: collectives of one
        process group pair up across ranks by the order they were issued in, so a bucket whose gradients
        happen to finish early waits for the buckets below it (and a rank with an empty shard, which issues
        0, 1, 2 after the fact, matches).  ``force``: backward is over, whatever has not fired will not."""
        n = len(self._waiting)
        while self._next_bucket < n and (force or self._waiting[self._next_bucket] <= 0):
            self.buckets.reduce(self._next_bucket)
            self._next_bucket += 1

    # ------------------------------------------------------------------ params
    def named_params(self) -> Dict[str, torch.Tensor]:
        return self.p

    def named_grads(self) -> Dict[str, torch.Tensor]:
        return self.g

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Keys as a reference GTVAE checkpoint names them."""
        return {k: v.detach().clone() for k, v in self.model.state_dict().items()}

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        with torch.no_grad():
            missing = [k for k in self.p if k not in sd]
            if missing:
                raise _lib.GtgError("state dict does not cover the parameters %s" % missing)
            for k, v in self.p.items():
                v.copy_(sd[k].to(self.dev, torch.float32))

    def deterministic(self, on: bool = True):
        """Training mode without its two random ingredients (dropout 0.5, word drop 0.3; text_vae.py:36,70):
        what parity tests and rehearsals run, since the stock LSTM backward exists in training mode only."""
        self.model.train()
        self.model.text_vae.drop.p = 0.0 if on else 0.5
        self.model.text_vae.word_drop = 0.0 if on else 0.3
        return self

    # ------------------------------------------------------------------ the step
    def _forward(self, batch: GraphBatch, dec_seq: torch.Tensor, eps: torch.Tensor, kl_w: float, report: bool = False):
        m = self.model
        res = m(None, batch.edge_index, batch.num_graphs, batch.adj_padded,
                batch.gold_bits if batch.gold_bits is not None else batch.gold_edges, dec_seq, kl_w, eps=eps,
                report=report)
        return {"loss": res[0], "text_acc": res[1], "kl": res[2], "graph_recon": res[3], "text_recon": res[4],
                "edge_recall": res[5], "edge_precision": res[6]}

    def step(self, batch: GraphBatch, dec_seq: torch.Tensor, eps: torch.Tensor, kl_w: float = 1.0, update: bool = True,
             global_graphs: Optional[int] = None):
        """Forward + backward (+ all-reduce + Adam when ``update``) on this rank's shard: ``batch`` on the
        device, ``dec_seq`` (B, T) token ids (pad = padidx), ``eps`` (B, latent) the latent noise.  The
        module's train()/eval() mode decides dropout and word noise.  Returns device scalars."""
        B = batch.num_graphs
        self.flat_g.zero_()
        self._waiting = list(self._expected)
        self._next_bucket = 0
        self._reduce_now = update and self.world > 1
        if global_graphs is None:
            global_graphs = batch.global_graphs
        if B > 0:
            res = self._forward(batch, dec_seq, eps, kl_w)
            w = shard_weight(B, global_graphs, self.world)
            (res["loss"] if w == 1.0 else res["loss"] * w).backward()
        else:
            zero = torch.zeros((), device=self.dev)
            res = {k: zero for k in ("loss", "text_acc", "kl", "graph_recon", "text_recon")}
        if update:
            if self.world > 1:
                self._launch_ready_buckets(force=True)   # empty shard, or a bucket whose hooks did not all fire
                self.buckets.finish()
            check(self.L.gtg_adam_step(ptr(self.flat_p), ptr(self.flat_g), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                       self.numel, ptr(self.adam_state), self.lr, self.betas[0], self.betas[1],
                                       self.eps, self.buckets.scale, torch.cuda.current_stream(self.dev).cuda_stream),
                  "gtg_adam_step")
        self._reduce_now = False
        return {k: (v.detach() if torch.is_tensor(v) else v) for k, v in res.items()}

    @torch.no_grad()
    def evaluate(self, batch: GraphBatch, dec_seq: torch.Tensor, eps: torch.Tensor, kl_w: float = 1.0,
                 report: bool = True):
        """The dev() pass (train.py:228-282): forward only; with ``report`` the spanning-tree read-out's edge
        recall / precision (means over the batch) as well.  Call ``model.eval()`` first, as dev() does."""
        return self._forward(batch, dec_seq, eps, kl_w, report=report)
