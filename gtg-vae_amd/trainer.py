"""Fused graph-VAE training step: forward, loss, backward and Adam as a fixed
sequence of C-ABI calls over preallocated device buffers (no autograd, no
allocation per step), optionally replayed as one HIP graph, data-parallel
over one process per GPU.

The step optimises  graph_recon + kl_w * KL  with joint_h := graph_h — the
reference's own pieces (graph_vae.py:22-73, gtvae.py:33-42, 107-113, 122-123)
composed with the reconstruction term that gtvae.py:124 leaves commented out,
and Adam(lr=1e-3, betas=(0.5, 0.999)) as train.py:290.  The module classes in
graph_vae.py / gtvae.py run the same kernels through autograd; this class is
the throughput path the benchmark and the training loop use.
"""
from typing import Dict, Optional

import torch

from . import _lib
from ._lib import check, ptr
from .data import GraphBatch
from .parallel import GradBuckets, shard_weight
from .streams import device_work_lock, owned_stream

# Captured steps kept per trainer, one per (batch size, update, edge capacity): an epoch whose last batch is
# short alternates between two shapes and must not re-capture either.  Oldest out beyond this many.
_MAX_GRAPHS = 16

# (state_dict key, shape as a function of (f, d, hid, m)); bucket 0 = decoder +
# latent head (gradients final before the encoder's backward starts), bucket 1 = encoder.
_PARAM_ORDER = (
    ("decoder.decode_1.weight", lambda f, d, h, m: (h, d)),
    ("decoder.decode_1.bias", lambda f, d, h, m: (h,)),
    ("decoder.decode_2.weight", lambda f, d, h, m: (m, h)),
    ("decoder.decode_2.bias", lambda f, d, h, m: (m,)),
    # the two latent weight matrices back to back: gtg_latent_bwd then multiplies by [Wmu; Wlv] in one GEMM
    ("zg_mu.weight", lambda f, d, h, m: (d, d)),
    ("zg_var.weight", lambda f, d, h, m: (d, d)),
    ("zg_mu.bias", lambda f, d, h, m: (d,)),
    ("zg_var.bias", lambda f, d, h, m: (d,)),
    ("encoder.conv1.lin.weight", lambda f, d, h, m: (2 * d, f)),
    ("encoder.conv1.bias", lambda f, d, h, m: (2 * d,)),
    ("encoder.conv2.lin.weight", lambda f, d, h, m: (d, 2 * d)),
    ("encoder.conv2.bias", lambda f, d, h, m: (d,)),
)
_FIRST_ENCODER = "encoder.conv1.lin.weight"


def _align(n: int, a: int = 64) -> int:
    return (n + a - 1) // a * a


class GraphVAETrainer:
    def __init__(self, input_dim: int, latent_dim: int, mlp_hid_dim: int, nmax: int, device="cuda",
                 init: Optional[Dict[str, torch.Tensor]] = None, lr: float = 1e-3, betas=(0.5, 0.999),
                 eps: float = 1e-8, max_batch: int = 0, max_edges: int = 0, group=None, seed: int = 0,
                 matrix_mode=None):
        if nmax > 64:
            raise _lib.GtgError("nmax > 64 is not supported")
        if input_dim != nmax:
            raise _lib.GtgError("the fused step feeds identity node features: input_dim must equal nmax")
        self.L = _lib.lib()
        self.dev = torch.device(device)
        if self.dev.type != "cuda":
            raise _lib.GtgError("GraphVAETrainer needs a GPU device; there is no CPU path")
        self.f, self.d, self.hid, self.n = input_dim, latent_dim, mlp_hid_dim, nmax
        self.m = nmax * (nmax + 1) // 2
        self.lr, self.betas, self.eps = lr, betas, eps
        # matrix pipe of the large GEMMs: "f32" | "f16x2" (the default) | "bf16x3" (both fp32-accurate) | "bf16"
        self.matrix_mode = _lib.matrix_mode(matrix_mode)

        # ---- flat parameter / gradient / moment buffers -----------------------
        self.offsets: Dict[str, tuple] = {}
        off = 0
        split = 0
        for name, shp in _PARAM_ORDER:
            if name == _FIRST_ENCODER:
                split = off
            shape = shp(self.f, self.d, self.hid, self.m)
            numel = 1
            for s in shape:
                numel *= s
            self.offsets[name] = (off, numel, shape)
            off = _align(off + numel)
        self.numel = off
        self.flat_p = torch.zeros(self.numel, dtype=torch.float32, device=self.dev)
        self.flat_g = torch.zeros_like(self.flat_p)
        self.exp_avg = torch.zeros_like(self.flat_p)
        self.exp_avg_sq = torch.zeros_like(self.flat_p)
        self.adam_state = torch.zeros(4, dtype=torch.int32, device=self.dev)
        self.p = {k: self.flat_p[o:o + n].view(s) for k, (o, n, s) in self.offsets.items()}
        self.g = {k: self.flat_g[o:o + n].view(s) for k, (o, n, s) in self.offsets.items()}
        self.buckets = GradBuckets(self.flat_g, [0, split, self.numel], group)
        self.world = self.buckets.world
        self._init_params(init, seed)

        # device scalars the captured step reads: [0] kl_w * shard weight (the KL term's factor in the backward),
        # [1] shard weight alone (the reconstruction gradient's factor).  The shard weight B_local * world /
        # B_global makes the sum all-reduce + 1/world of Adam the full-batch mean even when the ranks'
        # shards of a batch are unequal; it is 1 otherwise.
        self.kl_w_dev = torch.ones(2, dtype=torch.float32, device=self.dev)
        self._kl_w, self._shard_w = 1.0, 1.0
        self.scalars = torch.zeros(2, dtype=torch.float32, device=self.dev)  # [recon, kl]
        self._B = 0
        self._cap_B = 0
        self._cap_E = 0
        self._graphs: Dict[tuple, list] = {}
        self.captures = 0     # how many times a step has been captured (tests, diagnostics)
        if max_batch:
            self._ensure_capacity(max_batch, max_edges or max_batch * nmax * 4)

    # ------------------------------------------------------------------ params
    def _init_params(self, init, seed):
        if init is None:
            # the module classes carry the reference's initialisers
            from .gtvae import GTVAE

            torch.manual_seed(seed)
            model = GTVAE(self.f, self.d, self.hid, self.n, text=False)
            init = {k.replace("graph_vae.", ""): v for k, v in model.state_dict().items()}
        loaded = set()
        with torch.no_grad():
            for k, v in init.items():
                k = k.replace("graph_vae.", "")
                # torch_geometric 1.x checkpoints keep GCNConv's weight as `<conv>.weight` with shape (in, out)
                for conv in ("encoder.conv1", "encoder.conv2"):
                    if k == conv + ".weight":
                        k, v = conv + ".lin.weight", v.t()
                if k in self.p:
                    if tuple(v.shape) != tuple(self.p[k].shape):
                        raise _lib.GtgError("parameter %s has shape %s, expected %s" % (k, tuple(v.shape),
                                                                                          tuple(self.p[k].shape)))
                    self.p[k].copy_(v.to(self.dev, torch.float32))
                    loaded.add(k)
        missing = [k for k in self.p if k not in loaded]
        if missing:
            raise _lib.GtgError("state dict does not cover the parameters %s" % missing)

    def named_params(self) -> Dict[str, torch.Tensor]:
        return self.p

    def named_grads(self) -> Dict[str, torch.Tensor]:
        return self.g

    def state_dict(self) -> Dict[str, torch.Tensor]:
        """Keys as a reference GTVAE checkpoint names them."""
        out = {}
        for k, v in self.p.items():
            key = k if k.startswith("zg_") else "graph_vae." + k
            out[key] = v.detach().clone()
        return out

    def load_state_dict(self, sd: Dict[str, torch.Tensor]):
        self._init_params(sd, 0)

    # ------------------------------------------------------------------ buffers
    def _ensure_capacity(self, B: int, E: int):
        if B <= self._cap_B and E <= self._cap_E:
            return
        B = max(B, self._cap_B)
        E = max(E, self._cap_E)
        L, dev, n, d, hid, m, f = self.L, self.dev, self.n, self.d, self.hid, self.m, self.f
        f32 = dict(dtype=torch.float32, device=dev)
        self.ei = torch.zeros((2, E), dtype=torch.int64, device=dev)
        self.counts = torch.zeros(B * n * n, dtype=torch.int32, device=dev)  # zero in, zero out (gtg_adj_build)
        self.err = torch.zeros(1, dtype=torch.int32, device=dev)
        self.gather_err = torch.zeros(1, dtype=torch.int32, device=dev)
        self.gather_off = torch.zeros(B + 1, dtype=torch.int64, device=dev)
        self.adj = torch.empty((B, n, n), **f32)
        self.target = torch.empty((B, n, n), **f32)
        self.eps_buf = torch.empty((B, d), **f32)
        self.h1 = torch.empty((B * n + n, 2 * d), **f32)  # compact rows: R real + n prototype rows are valid
        self.meta = torch.zeros(2 * B + 2, dtype=torch.int32, device=dev)
        self.adjc = torch.empty((B * n + n, n), **f32)
        self.pooled = torch.empty((B, d), **f32)
        self.argmax = torch.empty((B, d), dtype=torch.int32, device=dev)
        self.mu = torch.empty((B, d), **f32)
        self.logvar = torch.empty((B, d), **f32)
        self.z = torch.empty((B, d), **f32)
        self.y1 = torch.empty((B, hid), **f32)
        self.out = torch.empty((B, m), **f32)
        self.dlogits = torch.empty((B, m), **f32)  # d(recon loss)/d(decoder logits), written by the loss kernel
        self.dz = torch.empty((B, d), **f32)
        self.dh = torch.empty((B, d), **f32)
        nbytes = max(L.gtg_encoder_compact_fwd_workspace_bytes(B, n, d), L.gtg_encoder_compact_bwd_workspace_bytes(B, n, d),
                     L.gtg_latent_workspace_bytes(B, d), L.gtg_decoder_bwd_workspace_bytes(B, d, hid, m),
                     L.gtg_bce_workspace_bytes(B), L.gtg_head_workspace_bytes(B, n, d, hid))
        self.ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        self._cap_B, self._cap_E = B, E
        self.drop_graphs()       # they hold the old buffers' addresses

    def drop_graphs(self):
        """Forget every captured step (after anything a capture froze has changed: buffers, matrix mode)."""
        self._graphs.clear()

    def load_batch(self, batch: GraphBatch, eps: torch.Tensor, global_graphs: Optional[int] = None):
        """Copy one batch into the step's static input buffers.  The edge list
        is padded to capacity with (0,0) self loops, which the adjacency kernel
        drops, so the launch shapes do not depend on the number of edges.
        ``global_graphs`` (default: the batch's own ``global_graphs`` field): size of the global batch
        this rank's shard was cut from, when the shards of the ranks may be unequal.  A shard of zero graphs
        is legal: the rank then contributes zero gradients and still joins the all-reduces."""
        B, E = batch.num_graphs, batch.edge_index.shape[1]
        if B == 0:
            self._B = 0
            return
        self._ensure_capacity(B, E)
        ei, adj = batch.edge_index, batch.adj_padded
        if (ei.is_cuda and adj.is_cuda and eps.is_cuda and ei.dtype == torch.int64 and ei.is_contiguous()
                and adj.dtype == torch.float32 and adj.is_contiguous() and eps.dtype == torch.float32
                and eps.is_contiguous()):
            check(self.L.gtg_stage_batch(ptr(ei), E, ptr(self.ei), self._cap_E, ptr(adj), ptr(self.target), adj.numel(),
                                         ptr(eps), ptr(self.eps_buf), eps.numel(),
                                         torch.cuda.current_stream(self.dev).cuda_stream), "gtg_stage_batch")
        else:  # host tensors or other dtypes: let torch convert and copy
            self.ei[:, :E].copy_(ei, non_blocking=True)
            if E < self._cap_E:
                self.ei[:, E:].zero_()
            self.target[:B].copy_(adj, non_blocking=True)
            self.eps_buf[:B].copy_(eps, non_blocking=True)
        self._B = B
        self._set_shard_weight(B, getattr(batch, "global_graphs", None) if global_graphs is None else global_graphs)

    def _set_shard_weight(self, B: int, global_graphs):
        w = shard_weight(B, global_graphs, self.world)
        if w != self._shard_w:
            self._shard_w = w
            self._push_weights()

    def _push_weights(self):
        self.kl_w_dev[0:1].fill_(self._kl_w * self._shard_w)   # stream-ordered fills: no host sync
        self.kl_w_dev[1:2].fill_(self._shard_w)

    def load_indices(self, dataset, graph_ids: torch.Tensor, eps: torch.Tensor, global_graphs: Optional[int] = None):
        """Assemble the batch of ``graph_ids`` (int64, on the device) from a ``data.DeviceDataset`` straight into
        the step's static buffers: two kernel launches, nothing read back.  ``global_graphs``: as load_batch
        (default: the ``global_graphs`` attribute DeviceLoader sets on the ids it yields)."""
        B = int(graph_ids.numel())
        if B == 0:
            self._B = 0
            return
        if global_graphs is None:
            global_graphs = getattr(graph_ids, "global_graphs", None)
        self._ensure_capacity(B, B * max(dataset.max_edges_per_graph, 1))
        if dataset.nmax != self.n:
            raise _lib.GtgError("dataset nmax %d != trainer nmax %d" % (dataset.nmax, self.n))
        ids = graph_ids if (graph_ids.dtype == torch.int64 and graph_ids.is_contiguous()) else graph_ids.long().contiguous()
        stream = torch.cuda.current_stream(self.dev).cuda_stream
        check(self.L.gtg_gather_batch(ptr(dataset.edges_local), dataset.total_edges, ptr(dataset.edge_start),
                                      ptr(dataset.adj_all), dataset.num_graphs, self.n, ptr(ids), B, ptr(self.gather_off),
                                      ptr(self.ei), self._cap_E, ptr(self.target), ptr(self.gather_err), stream),
              "gtg_gather_batch")
        self.eps_buf[:B].copy_(eps, non_blocking=True)
        self._B = B
        self._set_shard_weight(B, global_graphs)

    # ------------------------------------------------------------------ the step
    # The step is three runs of launches with the two gradient all-reduces between
    # them; each run can be replayed as a HIP graph while the collectives stay
    # ordinary stream work.
    def _seg_forward(self, B: int, stream: int, backward: bool = True, update: bool = False):
        """Forward pass; with ``backward`` also the backward of the decoder and the latent head (their
        gradients, bucket 0, are final after this), and with ``update`` the advance of Adam's step counter."""
        L, n, d, hid, m = self.L, self.n, self.d, self.hid, self.m
        p = self.p
        ws, wsn = ptr(self.ws), self.ws.numel()
        # meta is carved per call: [n_b (B) | offsets (B+1) | R + n] for THIS batch size; the adjacency
        # build leaves the graphs' extents n_b in its first B words
        meta = self.meta[: 2 * B + 2]
        check(L.gtg_adj_build(ptr(self.ei), self.ei.shape[1], B, n, ptr(self.counts), ptr(self.err), ptr(self.adj),
                              ptr(meta), stream), "gtg_adj_build")
        check(L.gtg_encoder_compact_fwd(ptr(self.adj), ptr(meta), ptr(p["encoder.conv1.lin.weight"]),
                                        ptr(p["encoder.conv1.bias"]), ptr(p["encoder.conv2.lin.weight"]),
                                        ptr(p["encoder.conv2.bias"]), B, n, d, ptr(meta), ptr(self.adjc), ptr(self.h1),
                                        ptr(self.pooled), ptr(self.argmax), self.matrix_mode, ws, wsn, stream),
              "gtg_encoder_compact_fwd")
        if backward:
            # latent head, decoder and loss, forward and backward, as one call: the loss terms' final sums and
            # Adam's step-counter kernel ride in a later launch of the call instead of being launches themselves
            g = self.g
            check(L.gtg_head_fwd_bwd(
                ptr(self.pooled), ptr(p["zg_mu.weight"]), ptr(p["zg_mu.bias"]), ptr(p["zg_var.weight"]),
                ptr(p["zg_var.bias"]), ptr(p["decoder.decode_1.weight"]), ptr(p["decoder.decode_1.bias"]),
                ptr(p["decoder.decode_2.weight"]), ptr(p["decoder.decode_2.bias"]), ptr(self.eps_buf), ptr(self.target),
                B, n, d, hid, ptr(self.kl_w_dev), self.kl_w_dev[1:].data_ptr(), ptr(self.mu), ptr(self.logvar),
                ptr(self.z), ptr(self.y1), ptr(self.out), ptr(self.dlogits), self.scalars.data_ptr(),
                self.scalars[1:].data_ptr(), ptr(g["zg_mu.weight"]), ptr(g["zg_mu.bias"]), ptr(g["zg_var.weight"]),
                ptr(g["zg_var.bias"]), ptr(g["decoder.decode_1.weight"]), ptr(g["decoder.decode_1.bias"]),
                ptr(g["decoder.decode_2.weight"]), ptr(g["decoder.decode_2.bias"]), ptr(self.dz), ptr(self.dh),
                ptr(self.adam_state) if update else None, self.lr, self.betas[0], self.betas[1], ws, wsn, stream),
                "gtg_head_fwd_bwd")
            return
        check(L.gtg_latent_fwd(ptr(self.pooled), ptr(p["zg_mu.weight"]), ptr(p["zg_mu.bias"]), ptr(p["zg_var.weight"]),
                               ptr(p["zg_var.bias"]), ptr(self.eps_buf), B, d, ptr(self.mu), ptr(self.logvar),
                               ptr(self.z), self.scalars[1:].data_ptr(), ws, wsn, stream), "gtg_latent_fwd")
        check(L.gtg_decoder_fwd(ptr(self.z), ptr(p["decoder.decode_1.weight"]), ptr(p["decoder.decode_1.bias"]),
                                ptr(p["decoder.decode_2.weight"]), ptr(p["decoder.decode_2.bias"]), B, d, hid, m,
                                ptr(self.y1), ptr(self.out), stream), "gtg_decoder_fwd")
        check(L.gtg_bce_loss_fwd(ptr(self.out), ptr(self.target), B, n, self.scalars.data_ptr(), None,
                                 self.kl_w_dev[1:].data_ptr(), ws, wsn, stream), "gtg_bce_loss_fwd")

    def _seg_encoder_backward(self, B: int, stream: int):
        L, n, d = self.L, self.n, self.d
        p, g = self.p, self.g
        meta = self.meta[: 2 * B + 2]
        check(L.gtg_encoder_compact_bwd(ptr(self.adj), ptr(self.adjc), ptr(meta), ptr(p["encoder.conv2.lin.weight"]),
                                        ptr(self.h1), ptr(self.argmax), ptr(self.dh), B, n, d,
                                        ptr(g["encoder.conv1.lin.weight"]), ptr(g["encoder.conv1.bias"]),
                                        ptr(g["encoder.conv2.lin.weight"]), ptr(g["encoder.conv2.bias"]), self.matrix_mode,
                                        ptr(self.ws), self.ws.numel(), stream), "gtg_encoder_compact_bwd")

    def _seg_adam(self, stream: int, prepared: bool = False):
        """The optimiser step; ``prepared``: this step's forward segment already advanced the step counter."""
        if prepared:
            check(self.L.gtg_adam_update(ptr(self.flat_p), ptr(self.flat_g), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                         self.numel, ptr(self.adam_state), self.betas[0], self.betas[1], self.eps,
                                         self.buckets.scale, stream), "gtg_adam_update")
            return
        check(self.L.gtg_adam_step(ptr(self.flat_p), ptr(self.flat_g), ptr(self.exp_avg), ptr(self.exp_avg_sq),
                                   self.numel, ptr(self.adam_state), self.lr, self.betas[0], self.betas[1], self.eps,
                                   self.buckets.scale, stream), "gtg_adam_step")

    def _segments(self, B: int, update: bool, backward: bool = True):
        """The step as a list of callables(stream), with None where an all-reduce goes."""
        segs = [lambda st: self._seg_forward(B, st, backward, backward and update)]
        if backward:
            segs += [None, lambda st: self._seg_encoder_backward(B, st)]
            if update:
                segs += [None, lambda st: self._seg_adam(st, prepared=True)]
        if self.world == 1:
            # no all-reduce to place between the pieces: one segment, hence one HIP graph per step
            parts = [f for f in segs if f is not None]
            return [lambda st: [f(st) for f in parts]]
        return segs

    def _run_segments(self, runners, update: bool):
        """Run the step's segments in order; the i-th gap between them carries the
        all-reduce of gradient bucket i (decoder + latent head first, so that it
        travels while the encoder's backward runs)."""
        bucket = 0
        last = len(runners) - 1
        for i, run in enumerate(runners):
            if run is None:
                if update and self.world > 1:
                    self.buckets.reduce(bucket)
                    if i == last - 1:   # the gap before the optimiser: all gradients must have arrived
                        self.buckets.finish()
                bucket += 1
            else:
                run()

    def _launch(self, B: int, update: bool, stream: int, backward: bool = True):
        segs = self._segments(B, update, backward)
        self._run_segments([None if f is None else (lambda f=f: f(stream)) for f in segs], update)

    def set_kl_weight(self, kl_w: float):
        if float(kl_w) != self._kl_w:
            self._kl_w = float(kl_w)
            self._push_weights()

    def run_loaded(self, update: bool = True, use_graph: bool = False):
        """One step on the batch already in the static buffers.  With
        ``use_graph`` the launches between the all-reduces are replayed as HIP
        graphs (captured on first use for this batch size)."""
        B = self._B
        if B == 0:
            # this rank's shard of the batch is empty: zero gradients, same collectives, same optimiser step
            self.flat_g.zero_()
            self.scalars.zero_()
            if update:
                self.buckets.reduce_all()
                self._seg_adam(torch.cuda.current_stream(self.dev).cuda_stream)
            return {"recon": self.scalars[0], "kl": self.scalars[1]}
        if use_graph:
            key = (B, update, self.ei.shape[1], self.matrix_mode)
            graphs = self._graphs.get(key)
            if graphs is None:
                graphs = self._capture(B, update)
                while len(self._graphs) >= _MAX_GRAPHS:
                    self._graphs.pop(next(iter(self._graphs)))
                self._graphs[key] = graphs
            self._run_segments([None if g is None else g.replay for g in graphs], update)
        else:
            self._launch(B, update, torch.cuda.current_stream(self.dev).cuda_stream)
        return {"recon": self.scalars[0], "kl": self.scalars[1]}

    def _capture(self, B: int, update: bool):
        """The step's segments as HIP graphs, captured on the package's own capture stream.

        That stream is created once per device with hipStreamCreateWithFlags (streams.owned_stream), so its
        handle is not one of PyTorch's 32 pooled streams: no ``torch.cuda.Stream()`` drawn elsewhere (a
        loader's copy stream, the collectives' streams) can turn out to be the stream under capture.  The
        whole capture, warm-up included, holds ``device_work_lock``, which the prefetching loader's thread
        takes around every pin -> copy -> record: no thread of this package issues device work meanwhile."""
        segs = self._segments(B, update)
        cur = torch.cuda.current_stream(self.dev)
        cap = owned_stream(self.dev, "capture")
        if cap.cuda_stream == cur.cuda_stream:
            raise _lib.GtgError("the step cannot be captured from inside its own capture stream")
        with device_work_lock:
            # warm up outside capture (no collectives: the optimiser state is rolled back below)
            snap = (self.flat_p.clone(), self.exp_avg.clone(), self.exp_avg_sq.clone(), self.adam_state.clone())
            cap.wait_stream(cur)
            with torch.cuda.stream(cap):
                for f in segs:
                    if f is not None:
                        f(cap.cuda_stream)
            graphs = []
            for f in segs:
                if f is None:
                    graphs.append(None)
                    continue
                g = torch.cuda.CUDAGraph()
                # thread_local: threads this package does not own (the collectives' watchdog, a caller's own
                # loader) may call the runtime meanwhile without invalidating the capture
                with torch.cuda.graph(g, stream=cap, capture_error_mode="thread_local"):
                    f(cap.cuda_stream)
                graphs.append(g)
            cur.wait_stream(cap)
            self.flat_p.copy_(snap[0]); self.exp_avg.copy_(snap[1]); self.exp_avg_sq.copy_(snap[2])
            self.adam_state.copy_(snap[3])
        self.captures += 1
        return graphs

    def step(self, batch: GraphBatch, eps: torch.Tensor, kl_w: float = 1.0, update: bool = True,
             use_graph: bool = False):
        """Forward + backward (+ all-reduce + Adam when ``update``) on one batch.
        Returns device scalars {"recon", "kl"}; ``last_out`` holds the decoder
        probabilities (B, M) of this step."""
        self.set_kl_weight(kl_w)
        self.load_batch(batch, eps)
        res = self.run_loaded(update, use_graph)
        self.last_out = self.out[: self._B]
        return res

    @torch.no_grad()
    def evaluate(self, batch: GraphBatch, eps: torch.Tensor, report: bool = True):
        """Forward only (the dev() pass of train.py:228-282): reconstruction loss,
        KL and, with ``report``, the batch means of edge recall / precision of
        the maximum-spanning-tree read-out against the gold edges."""
        from . import functional as Fn
        from .maxspantree import mst_graph_batched

        self.load_batch(batch, eps)
        self._launch(self._B, False, torch.cuda.current_stream(self.dev).cuda_stream, backward=False)
        self.last_out = self.out[: self._B]
        res = {"recon": self.scalars[0].clone(), "kl": self.scalars[1].clone()}
        if report:
            edges, count = mst_graph_batched(self.last_out, self.n)
            rec, prec = Fn.edge_precision_recall(edges, count, batch.gold_bits, self.n)
            res["edge_recall"], res["edge_precision"] = rec.mean(), prec.mean()
            res["edges"], res["count"] = edges, count
        return res

    def edges_ok(self) -> bool:
        """Host read-back of the adjacency and batch-gather kernels' error flags (a sync).  The adjacency
        flag is sticky on the device (the library only ever sets it): it is cleared here once reported."""
        bad_edges = int(self.err.item()) != 0
        if bad_edges:
            self.err.zero_()
        return not bad_edges and int(self.gather_err.item()) == 0
