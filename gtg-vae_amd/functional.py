"""torch.autograd.Function wrappers over the C ABI (include/gtgvae.h).

PyTorch is used for device memory, streams and the autograd graph only; every
arithmetic step of forward and backward runs in csrc/*.hip.  All tensors must
be fp32 and on the GPU; there is no fallback.
"""
import functools
from typing import Optional

import torch

from . import _lib
from ._lib import check, ptr, stream_ptr


def _on_tensor_device(fn):
    """Run ``fn`` with the device of its first GPU tensor argument current, so that the stream handed to the
    library (the current stream) and the kernels' launch device are the tensors' device even when the caller's
    current device is another GPU."""
    @functools.wraps(fn)
    def wrapped(*args, **kwargs):
        dev = next((a.device for a in args if torch.is_tensor(a) and a.is_cuda), None)
        if dev is None or dev.index == torch.cuda.current_device():
            return fn(*args, **kwargs)
        with torch.cuda.device(dev):
            return fn(*args, **kwargs)
    return wrapped


def _f32c(t: Optional[torch.Tensor]) -> Optional[torch.Tensor]:
    if t is None:
        return None
    _lib.require_cuda(t)
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty((max(int(nbytes), 256),), dtype=torch.uint8, device=device)


# ---------------------------------------------------------------------------
# adjacency
# ---------------------------------------------------------------------------
@_on_tensor_device
def build_adj(edge_index: torch.Tensor, num_graphs: int, nmax: int, check_edges: bool = False,
              return_extents: bool = False):
    """D^-1/2 (C + I) D^-1/2 per graph, (B, nmax, nmax) fp32, from a PyG-style
    batched edge_index (2, E) int64.  With check_edges the error flag is read
    back (a host sync) and an exception raised if an edge leaves its graph.
    With return_extents the per-graph extents (B) int32 are returned as well
    (1 + the highest node index an edge touches: the rows the compact encoder
    stores; EncoderFn's ``extents`` argument)."""
    _lib.require_cuda(edge_index)
    ei = edge_index.contiguous()
    if ei.dtype != torch.int64:
        ei = ei.long()
    dev = ei.device
    counts = torch.zeros((num_graphs * nmax * nmax,), dtype=torch.int32, device=dev)  # zero in, zero out
    err = torch.zeros((1,), dtype=torch.int32, device=dev)
    adj = torch.empty((num_graphs, nmax, nmax), dtype=torch.float32, device=dev)
    extents = torch.empty((num_graphs,), dtype=torch.int32, device=dev)
    check(_lib.lib().gtg_adj_build(ptr(ei), ei.shape[1], num_graphs, nmax, ptr(counts), ptr(err), ptr(adj),
                                   ptr(extents), stream_ptr()), "gtg_adj_build")
    if check_edges and int(err.item()) != 0:
        raise _lib.GtgError("edge_index holds an edge whose endpoints are out of range or in different graphs")
    return (adj, extents) if return_extents else adj


# ---------------------------------------------------------------------------
# encoder
# ---------------------------------------------------------------------------
class EncoderFn(torch.autograd.Function):
    """pooled (B, d) = global_max_pool(GCN2(relu(GCN1(x))))."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, adj, x, W1, b1, W2, b2, extents=None, matrix_mode=None):
        L = _lib.lib()
        ctx.mode = mode = _lib.matrix_mode(matrix_mode)
        adj, x, W1, b1, W2, b2 = map(_f32c, (adj, x, W1, b1, W2, b2))
        B, n, _ = adj.shape
        H, f = W1.shape
        d = W2.shape[0]
        if H != 2 * d or W2.shape[1] != H:
            raise _lib.GtgError("encoder weights must be W1 (2d, f), W2 (d, 2d)")
        if x is not None and tuple(x.shape) != (B * n, f):
            raise _lib.GtgError(f"x must be (B*n, f) = ({B * n}, {f}), got {tuple(x.shape)}")
        dev = adj.device
        ctx.compact = x is None and extents is not None and f == n
        if ctx.compact:
            # identity features: only the rows an edge touches are stored and multiplied
            meta = torch.empty((2 * B + 2,), dtype=torch.int32, device=dev)
            adjc = torch.empty((B * n + n, n), dtype=torch.float32, device=dev)
            h1c = torch.empty((B * n + n, H), dtype=torch.float32, device=dev)
            pooled = torch.empty((B, d), dtype=torch.float32, device=dev)
            argmax = torch.empty((B, d), dtype=torch.int32, device=dev)
            ws = _ws(L.gtg_encoder_compact_fwd_workspace_bytes(B, n, d), dev)
            check(L.gtg_encoder_compact_fwd(ptr(adj), ptr(extents), ptr(W1), ptr(b1), ptr(W2), ptr(b2), B, n, d,
                                            ptr(meta), ptr(adjc), ptr(h1c), ptr(pooled), ptr(argmax), mode, ptr(ws),
                                            ws.numel(), stream_ptr()), "gtg_encoder_compact_fwd")
            ctx.save_for_backward(adj, adjc, meta, W2, h1c, argmax)
            ctx.dims = (B, n, f, d)
            ctx.mark_non_differentiable(argmax)
            return pooled, argmax
        h1 = torch.empty((B * n, H), dtype=torch.float32, device=dev)
        pooled = torch.empty((B, d), dtype=torch.float32, device=dev)
        argmax = torch.empty((B, d), dtype=torch.int32, device=dev)
        nb = L.gtg_encoder_fwd_workspace_bytes(B, n, f, d, int(x is not None))
        ws = _ws(nb, dev)
        check(L.gtg_encoder_fwd(ptr(adj), ptr(x), ptr(W1), ptr(b1), ptr(W2), ptr(b2), B, n, f, d, ptr(h1), ptr(pooled),
                                ptr(argmax), mode, ptr(ws), ws.numel(), stream_ptr()), "gtg_encoder_fwd")
        ctx.save_for_backward(adj, x, W1, W2, h1, argmax)
        ctx.dims = (B, n, f, d)
        ctx.mark_non_differentiable(argmax)
        return pooled, argmax

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dpooled, _dargmax):
        L = _lib.lib()
        B, n, f, d = ctx.dims
        dpooled = _f32c(dpooled)
        dev = dpooled.device
        dW1 = torch.empty((2 * d, f), dtype=torch.float32, device=dev)
        db1 = torch.empty((2 * d,), dtype=torch.float32, device=dev)
        dW2 = torch.empty((d, 2 * d), dtype=torch.float32, device=dev)
        db2 = torch.empty((d,), dtype=torch.float32, device=dev)
        if ctx.compact:
            adj, adjc, meta, W2, h1c, argmax = ctx.saved_tensors
            ws = _ws(L.gtg_encoder_compact_bwd_workspace_bytes(B, n, d), dev)
            check(L.gtg_encoder_compact_bwd(ptr(adj), ptr(adjc), ptr(meta), ptr(W2), ptr(h1c), ptr(argmax),
                                            ptr(dpooled), B, n, d, ptr(dW1), ptr(db1), ptr(dW2), ptr(db2), ctx.mode,
                                            ptr(ws), ws.numel(), stream_ptr()), "gtg_encoder_compact_bwd")
            return None, None, dW1, db1, dW2, db2, None, None
        adj, x, W1, W2, h1, argmax = ctx.saved_tensors
        nb = L.gtg_encoder_bwd_workspace_bytes(B, n, f, d, int(x is not None))
        ws = _ws(nb, dev)
        check(L.gtg_encoder_bwd(ptr(adj), ptr(x), ptr(W1), ptr(W2), ptr(h1), ptr(argmax), ptr(dpooled), B, n, f, d,
                                ptr(dW1), ptr(db1), ptr(dW2), ptr(db2), ctx.mode, ptr(ws), ws.numel(), stream_ptr()),
              "gtg_encoder_bwd")
        # node features and the adjacency are inputs, not parameters: no gradient
        return None, None, dW1, db1, dW2, db2, None, None


# ---------------------------------------------------------------------------
# latent head
# ---------------------------------------------------------------------------
class LatentFn(torch.autograd.Function):
    """(z, kl, mu, logvar) from h with eps supplied; mu/logvar are returned for
    inspection only (non-differentiable outputs)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, h, Wmu, bmu, Wlv, blv, eps):
        L = _lib.lib()
        h, Wmu, bmu, Wlv, blv, eps = map(_f32c, (h, Wmu, bmu, Wlv, blv, eps))
        B, d = h.shape
        dev = h.device
        mu = torch.empty((B, d), dtype=torch.float32, device=dev)
        lv = torch.empty_like(mu)
        z = torch.empty_like(mu)
        kl = torch.empty((), dtype=torch.float32, device=dev)
        ws = _ws(L.gtg_latent_workspace_bytes(B, d), dev)
        check(L.gtg_latent_fwd(ptr(h), ptr(Wmu), ptr(bmu), ptr(Wlv), ptr(blv), ptr(eps), B, d, ptr(mu), ptr(lv), ptr(z),
                               ptr(kl), ptr(ws), ws.numel(), stream_ptr()), "gtg_latent_fwd")
        ctx.save_for_backward(h, Wmu, Wlv, mu, lv, eps)
        ctx.mark_non_differentiable(mu, lv)
        return z, kl, mu, lv

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dz, dkl, _dmu, _dlv):
        L = _lib.lib()
        h, Wmu, Wlv, mu, lv, eps = ctx.saved_tensors
        B, d = h.shape
        dev = h.device
        dz = _f32c(dz)
        dkl = _f32c(dkl) if dkl is not None else None
        dWmu = torch.empty_like(Wmu)
        dWlv = torch.empty_like(Wlv)
        dbmu = torch.empty((d,), dtype=torch.float32, device=dev)
        dblv = torch.empty((d,), dtype=torch.float32, device=dev)
        dh = torch.empty_like(h)
        ws = _ws(L.gtg_latent_workspace_bytes(B, d), dev)
        kl_w = 1.0 if dkl is not None else 0.0
        check(L.gtg_latent_bwd(ptr(h), ptr(Wmu), ptr(Wlv), ptr(mu), ptr(lv), ptr(eps), ptr(dz), kl_w, ptr(dkl), B, d,
                               ptr(dWmu), ptr(dbmu), ptr(dWlv), ptr(dblv), ptr(dh), ptr(ws), ws.numel(), stream_ptr()),
              "gtg_latent_bwd")
        return dh, dWmu, dbmu, dWlv, dblv, None


class ReparamKLFn(torch.autograd.Function):
    """(z, kl) from (mu, logvar, eps): GTVAE.reparameterize + GTVAE.loss_kl."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, mu, logvar, eps):
        L = _lib.lib()
        mu, logvar, eps = map(_f32c, (mu, logvar, eps))
        B, d = mu.shape
        z = torch.empty_like(mu)
        kl = torch.empty((), dtype=torch.float32, device=mu.device)
        ws = _ws(L.gtg_latent_workspace_bytes(B, d), mu.device)
        check(L.gtg_reparam_kl_fwd(ptr(mu), ptr(logvar), ptr(eps), B, d, ptr(z), ptr(kl), ptr(ws), ws.numel(),
                                   stream_ptr()), "gtg_reparam_kl_fwd")
        ctx.save_for_backward(mu, logvar, eps)
        return z, kl

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dz, dkl):
        L = _lib.lib()
        mu, logvar, eps = ctx.saved_tensors
        B, d = mu.shape
        dz = _f32c(dz)
        dkl = _f32c(dkl) if dkl is not None else None
        dmu = torch.empty_like(mu)
        dlv = torch.empty_like(mu)
        check(L.gtg_reparam_kl_bwd(ptr(mu), ptr(logvar), ptr(eps), ptr(dz), 1.0 if dkl is not None else 0.0, ptr(dkl),
                                   B, d, ptr(dmu), ptr(dlv), stream_ptr()), "gtg_reparam_kl_bwd")
        return dmu, dlv, None


# ---------------------------------------------------------------------------
# decoder and reconstruction loss
# ---------------------------------------------------------------------------
class DecoderFn(torch.autograd.Function):
    """out (B, M) = sigmoid(relu(z D1^T + c1) D2^T + c2)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, z, D1, c1, D2, c2):
        L = _lib.lib()
        z, D1, c1, D2, c2 = map(_f32c, (z, D1, c1, D2, c2))
        B, d = z.shape
        hid = D1.shape[0]
        m_out = D2.shape[0]
        y1 = torch.empty((B, hid), dtype=torch.float32, device=z.device)
        out = torch.empty((B, m_out), dtype=torch.float32, device=z.device)
        check(L.gtg_decoder_fwd(ptr(z), ptr(D1), ptr(c1), ptr(D2), ptr(c2), B, d, hid, m_out, ptr(y1), ptr(out),
                                stream_ptr()), "gtg_decoder_fwd")
        ctx.save_for_backward(z, D1, D2, y1, out)
        return out

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dout):
        L = _lib.lib()
        z, D1, D2, y1, out = ctx.saved_tensors
        B, d = z.shape
        hid, m_out = D1.shape[0], D2.shape[0]
        dev = z.device
        dout = _f32c(dout)
        dD1 = torch.empty_like(D1)
        dD2 = torch.empty_like(D2)
        dc1 = torch.empty((hid,), dtype=torch.float32, device=dev)
        dc2 = torch.empty((m_out,), dtype=torch.float32, device=dev)
        dz = torch.empty_like(z)
        ws = _ws(L.gtg_decoder_bwd_workspace_bytes(B, d, hid, m_out), dev)
        check(L.gtg_decoder_bwd(ptr(z), ptr(D1), ptr(D2), ptr(y1), ptr(out), ptr(dout), None, 0, 1.0, None, B, d, hid,
                                m_out, ptr(dD1), ptr(dc1), ptr(dD2), ptr(dc2), ptr(dz), ptr(ws), ws.numel(),
                                stream_ptr()), "gtg_decoder_bwd")
        return dz, dD1, dc1, dD2, dc2


class BCELossFn(torch.autograd.Function):
    """mean_b mean_k BCE(out[b,k], triu(adj[b])[k]) (log clamped at -100)."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, out, adj):
        L = _lib.lib()
        out, adj = _f32c(out), _f32c(adj)
        B, m = out.shape
        n = adj.shape[-1]
        if tuple(adj.shape) != (B, n, n) or m != n * (n + 1) // 2:
            raise _lib.GtgError(f"out {tuple(out.shape)} / adj {tuple(adj.shape)} do not match (B, n(n+1)/2) / (B, n, n)")
        loss = torch.empty((), dtype=torch.float32, device=out.device)
        ws = _ws(L.gtg_bce_workspace_bytes(B), out.device)
        check(L.gtg_bce_loss_fwd(ptr(out), ptr(adj), B, n, ptr(loss), None, None, ptr(ws), ws.numel(), stream_ptr()),
              "gtg_bce_loss_fwd")
        ctx.save_for_backward(out, adj)
        return loss

    @staticmethod
    @_on_tensor_device
    def backward(ctx, g):
        L = _lib.lib()
        out, adj = ctx.saved_tensors
        B, n = adj.shape[0], adj.shape[-1]
        g = _f32c(g)
        dout = torch.empty_like(out)
        check(L.gtg_bce_loss_bwd(ptr(out), ptr(adj), 1.0, ptr(g), B, n, 0, ptr(dout), stream_ptr()),
              "gtg_bce_loss_bwd")
        return dout, None


@_on_tensor_device
def edge_precision_recall(edges: torch.Tensor, count: torch.Tensor, gold_bits: torch.Tensor, nmax: int):
    """Per-graph (recall, precision) fp32 tensors on the GPU."""
    _lib.require_cuda(edges, count, gold_bits)
    G = count.shape[0]
    rec = torch.empty((G,), dtype=torch.float32, device=edges.device)
    prec = torch.empty_like(rec)
    check(_lib.lib().gtg_edge_prf(ptr(edges.contiguous()), ptr(count.contiguous()), ptr(gold_bits.contiguous()), G,
                                  nmax, ptr(rec), ptr(prec), stream_ptr()), "gtg_edge_prf")
    return rec, prec


@_on_tensor_device
def gemm(A, B, a_kc=True, b_kc=True, bias=None, epilogue=0, aux=None, out=None, M=None, N=None, K=None,
         split_k=True, matrix_mode="f32"):
    """Thin test/profiling hook over gtg_gemm_f32 (see include/gtgvae.h)."""
    L = _lib.lib()
    A, B = _f32c(A), _f32c(B)
    if M is None:
        M, K = (A.shape[0], A.shape[1]) if a_kc else (A.shape[1], A.shape[0])
        N = B.shape[0] if b_kc else B.shape[1]
    C = out if out is not None else torch.empty((M, N), dtype=torch.float32, device=A.device)
    slabs = None
    nb = L.gtg_gemm_splitk_workspace_bytes(M, N, K) if split_k else 0
    if nb:
        slabs = _ws(nb, A.device)
    check(L.gtg_gemm_f32(ptr(A), ptr(B), ptr(C), M, N, K, A.stride(0), B.stride(0), C.stride(0), int(a_kc), int(b_kc),
                         ptr(_f32c(bias)), epilogue, ptr(_f32c(aux)), aux.stride(0) if aux is not None else 0,
                         ptr(slabs), nb, _lib.matrix_mode(matrix_mode), stream_ptr()), "gtg_gemm_f32")
    return C


@_on_tensor_device
def gemm_rows(A, B, a_kc=True, b_kc=True, rows_dev=None, bias=None, epilogue=0, aux=None, out=None, matrix_mode="f32"):
    """Test/profiling hook over gtg_gemm_f32_rows: the GEMM of the compact-row
    encoder (row count read on the device, tail tiles split along K)."""
    L = _lib.lib()
    A, B = _f32c(A), _f32c(B)
    M, K = (A.shape[0], A.shape[1]) if a_kc else (A.shape[1], A.shape[0])
    N = B.shape[0] if b_kc else B.shape[1]
    C = out if out is not None else torch.zeros((M, N), dtype=torch.float32, device=A.device)
    ws = _ws(L.gtg_gemm_rows_workspace_bytes(), A.device)
    check(L.gtg_gemm_f32_rows(ptr(A), ptr(B), ptr(C), M, N, K, A.stride(0), B.stride(0), C.stride(0), int(a_kc),
                              int(b_kc), ptr(_f32c(bias)), epilogue, ptr(_f32c(aux)),
                              aux.stride(0) if aux is not None else 0, ptr(rows_dev), _lib.matrix_mode(matrix_mode),
                              ptr(ws), ws.numel(), stream_ptr()),
          "gtg_gemm_f32_rows")
    return C


@_on_tensor_device
def split_planes(x: torch.Tensor, transpose: bool = False) -> torch.Tensor:
    """k-blocked bf16 planes (3, K/16, R, 16) of the (R, K) operand x -- or of x^T with transpose -- whose sum is
    the operand exactly; K must be a multiple of 16."""
    x = _f32c(x)
    rows, cols = x.shape
    R, K = (cols, rows) if transpose else (rows, cols)
    out = torch.empty((3, K // 16, R, 16), dtype=torch.bfloat16, device=x.device)
    check(_lib.lib().gtg_split_planes(ptr(x), rows, cols, x.stride(0), int(transpose), ptr(out), stream_ptr()),
          "gtg_split_planes")
    return out


@_on_tensor_device
def gemm_planes_rows(Ap, Bp, rows_dev=None, bias=None, epilogue=0, aux=None, out=None):
    """Test/profiling hook over gtg_gemm_planes_rows: C = epilogue(A B^T) from k-blocked planes (split_planes)."""
    L = _lib.lib()
    _, KB, M, _ = Ap.shape
    K, N = KB * 16, Bp.shape[2]
    C = out if out is not None else torch.zeros((M, N), dtype=torch.float32, device=Ap.device)
    ws = _ws(L.gtg_gemm_rows_workspace_bytes(), Ap.device)
    check(L.gtg_gemm_planes_rows(ptr(Ap), ptr(Bp), ptr(C), M, N, K, C.stride(0), ptr(_f32c(bias)), epilogue,
                                 ptr(_f32c(aux)), aux.stride(0) if aux is not None else 0, ptr(rows_dev), ptr(ws),
                                 ws.numel(), stream_ptr()), "gtg_gemm_planes_rows")
    return C


# ---------------------------------------------------------------------------
# weight gradients written where the optimiser reads them
# ---------------------------------------------------------------------------
def _grad_slot(weight, shape) -> Optional[torch.Tensor]:
    """``weight.grad`` when the weight asks for its gradient to be STORED there: its ``grad_overwrite`` attribute
    is true and ``.grad`` is already a dense contiguous fp32 GPU tensor of ``shape`` (JointTrainer's flat gradient
    buffer).  The backward then writes the gradient into that tensor, replacing what it held -- right for a weight
    that one call per backward pass uses, in a buffer nobody accumulates over passes -- and returns None to
    autograd: no temporary, no ``grad += new`` pass over the weight.  (The weight's post-accumulate-grad hooks run
    as usual: autograd calls them for a None gradient too, after the backward that stored it has returned.)"""
    if weight is None or not getattr(weight, "grad_overwrite", False):
        return None
    g = weight.grad
    if g is not None and g.is_cuda and g.dtype == torch.float32 and g.is_contiguous() and tuple(g.shape) == tuple(shape):
        return g
    return None


# ---------------------------------------------------------------------------
# nn.Linear on the library's GEMM (the text branch's small layers)
# ---------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y (rows, out) = x W^T + b (text_vae.py:41-42 hgen, z2emb): forward, dx = dy W and dW = dy^T x on the fp32
    matrix pipe (the layers are a few hundred MFLOP: the matrix mode's split products would buy nothing); db, the
    column sums of dy, is a torch reduction."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, x, W, b):
        x2, Wc, bc = _f32c(x.reshape(-1, x.shape[-1])), _f32c(W), _f32c(b)
        if Wc.dim() != 2 or Wc.shape[1] != x2.shape[1] or (bc is not None and bc.numel() != Wc.shape[0]):
            raise _lib.GtgError("linear: x (..., in), W (out, in), b (out)")
        y = gemm(x2, Wc, True, True, bias=bc)
        ctx.save_for_backward(x2, Wc)
        ctx.lead = tuple(x.shape[:-1])
        ctx.has_bias = b is not None
        return y.view(ctx.lead + (Wc.shape[0],))

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dy):
        x2, W = ctx.saved_tensors
        dy2 = _f32c(dy.reshape(-1, W.shape[0]))
        dx = gemm(dy2, W, True, False).view(ctx.lead + (W.shape[1],)) if ctx.needs_input_grad[0] else None
        dW = gemm(dy2, x2, False, False) if ctx.needs_input_grad[1] else None
        db = dy2.sum(0) if (ctx.has_bias and ctx.needs_input_grad[2]) else None
        return dx, dW, db


def linear(x, W, b=None):
    """x W^T + b; see LinearFn."""
    return LinearFn.apply(x, W, b)


# ---------------------------------------------------------------------------
# vocabulary projection + token cross-entropy (text decoder)
# ---------------------------------------------------------------------------
class ProjCEFn(torch.autograd.Function):
    """loss (rows) = cross_entropy(H W^T + b, targets, ignore_index=padidx, reduction='none'),
    pred (rows) int32 = argmax of the logits (-1 where the target is the pad symbol).

    text_vae.py:65 (self.proj) + :77-79 (loss_rec) + :83-84 (accuracy's arg-max)
    without ever holding the logits of pad tokens: those rows are left out of
    all three GEMMs."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, H, W, b, targets, padidx, matrix_mode=None):
        L = _lib.lib()
        ctx.mode = mode = _lib.matrix_mode(matrix_mode)
        ctx.w_param = W
        H, W, b = map(_f32c, (H, W, b))
        _lib.require_cuda(targets)
        targets = targets.contiguous().long().view(-1)
        R, K = H.shape
        V = W.shape[0]
        if W.shape[1] != K or targets.numel() != R or (b is not None and b.numel() != V):
            raise _lib.GtgError("proj_ce: H (rows, k), W (vocab, k), b (vocab), targets (rows)")
        dev = H.device
        ldp = (V + 31) // 32 * 32
        loss = torch.empty((R,), dtype=torch.float32, device=dev)
        pred = torch.empty((R,), dtype=torch.int32, device=dev)
        rowmap = torch.empty((2 * R + 1,), dtype=torch.int32, device=dev)
        Hc = torch.empty((R, K), dtype=torch.float32, device=dev)
        P = torch.empty((R, ldp), dtype=torch.float32, device=dev)
        ws = _ws(L.gtg_proj_ce_fwd_workspace_bytes(R, K, V), dev)
        check(L.gtg_proj_ce_fwd(ptr(H), ptr(W), ptr(b), ptr(targets), R, K, V, int(padidx), ptr(loss), ptr(pred),
                                ptr(rowmap), ptr(Hc), ptr(P), ldp, mode, ptr(ws), ws.numel(), stream_ptr()),
              "gtg_proj_ce_fwd")
        ctx.save_for_backward(W, rowmap, Hc, P)
        ctx.dims = (R, K, V, ldp, b is not None)
        ctx.mark_non_differentiable(pred)
        return loss, pred

    @staticmethod
    @_on_tensor_device
    def backward(ctx, gloss, _gpred):
        L = _lib.lib()
        W, rowmap, Hc, P = ctx.saved_tensors
        R, K, V, ldp, has_bias = ctx.dims
        gloss = _f32c(gloss).view(-1)
        dev = gloss.device
        dH = torch.empty((R, K), dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        slot = _grad_slot(ctx.w_param, (V, K))   # the weight's own .grad, if it asks for that (see _grad_slot)
        dW = slot if slot is not None else torch.empty((V, K), dtype=torch.float32, device=dev)
        db = torch.empty((V,), dtype=torch.float32, device=dev) if has_bias else None
        ws = _ws(L.gtg_proj_ce_bwd_workspace_bytes(R, K, V), dev)
        check(L.gtg_proj_ce_bwd(ptr(W), ptr(gloss), ptr(rowmap), ptr(Hc), ptr(P), ldp, R, K, V, ptr(dH), ptr(dW),
                                ptr(db), ctx.mode, ptr(ws), ws.numel(), stream_ptr()), "gtg_proj_ce_bwd")
        return dH, (None if slot is not None else dW), db, None, None, None


def lstm_applies(T: int, B: int, I: int, H: int) -> bool:
    """Sizes the HIP recurrence takes (include/gtgvae.h, gtg_lstm_fwd)."""
    return T >= 1 and B >= 1 and I >= 4 and I % 4 == 0 and H >= 64 and H % 64 == 0 and T * B * 4 * H < 2 ** 31


class LSTMFn(torch.autograd.Function):
    """One direction of a one-layer nn.LSTM over a padded sequence (text_vae.py:37-40, called at :57 and :64):
    x (T, B, I) -> (every step's hidden state (T, B, H) in time order, h_n (B, H), c_n (B, H)).  ``reverse`` runs
    time backwards (the module's ``*_reverse`` weights); h0 / c0 (B, H) or None."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, x, w_ih, w_hh, b_ih, b_hh, h0, c0, reverse=False, matrix_mode=None):
        L = _lib.lib()
        ctx.mode = mode = _lib.matrix_mode(matrix_mode)
        ctx.w_params = (w_ih, w_hh)
        x, w_ih, w_hh, b_ih, b_hh, h0, c0 = map(_f32c, (x, w_ih, w_hh, b_ih, b_hh, h0, c0))
        T, B, I = x.shape
        H = w_hh.shape[1]
        if w_ih.shape != (4 * H, I) or w_hh.shape != (4 * H, H) or b_ih.numel() != 4 * H or b_hh.numel() != 4 * H or \
                any(t is not None and t.shape != (B, H) for t in (h0, c0)):
            raise _lib.GtgError("lstm: x (T, B, I), w_ih (4H, I), w_hh (4H, H), biases (4H), h0 / c0 (B, H)")
        if not lstm_applies(T, B, I, H):
            raise _lib.GtgError("lstm: the HIP recurrence needs I % 4 == 0 and H % 64 == 0")
        dev = x.device
        hs = torch.empty((T, B, H), dtype=torch.float32, device=dev)
        cs = torch.empty((T, B, H), dtype=torch.float32, device=dev)
        gates = torch.empty((T, B, 4 * H), dtype=torch.float32, device=dev)
        ws = _ws(L.gtg_lstm_fwd_workspace_bytes(T, B, I, H), dev)
        check(L.gtg_lstm_fwd(ptr(x), ptr(w_ih), ptr(w_hh), ptr(b_ih), ptr(b_hh), ptr(h0), ptr(c0), T, B, I, H,
                             int(bool(reverse)), ptr(hs), ptr(cs), ptr(gates), mode, ptr(ws), ws.numel(), stream_ptr()),
              "gtg_lstm_fwd")
        ctx.save_for_backward(x, w_ih, w_hh, h0, c0, hs, cs, gates)
        ctx.dims = (T, B, I, H, bool(reverse))
        # an output nobody differentiates through (the encoder uses h_n alone) reaches backward as None, which
        # gtg_lstm_bwd takes as "no gradient", instead of a (T, B, H) block of zeros read at every time step
        ctx.set_materialize_grads(False)
        last = 0 if reverse else T - 1
        return hs, hs[last].clone(), cs[last].clone()

    @staticmethod
    @_on_tensor_device
    def backward(ctx, d_hs, d_hT, d_cT):
        L = _lib.lib()
        x, w_ih, w_hh, h0, c0, hs, cs, gates = ctx.saved_tensors
        T, B, I, H, reverse = ctx.dims
        d_hs, d_hT, d_cT = map(_f32c, (d_hs, d_hT, d_cT))
        dev = x.device
        need = ctx.needs_input_grad
        dx = torch.empty((T, B, I), dtype=torch.float32, device=dev) if need[0] else None
        s_ih, s_hh = _grad_slot(ctx.w_params[0], (4 * H, I)), _grad_slot(ctx.w_params[1], (4 * H, H))
        dw_ih = s_ih if s_ih is not None else torch.empty((4 * H, I), dtype=torch.float32, device=dev)
        dw_hh = s_hh if s_hh is not None else torch.empty((4 * H, H), dtype=torch.float32, device=dev)
        db = torch.empty((4 * H,), dtype=torch.float32, device=dev)
        dh0 = torch.empty((B, H), dtype=torch.float32, device=dev) if (h0 is not None and need[5]) else None
        dc0 = torch.empty((B, H), dtype=torch.float32, device=dev) if (c0 is not None and need[6]) else None
        ws = _ws(L.gtg_lstm_bwd_workspace_bytes(T, B, I, H), dev)
        check(L.gtg_lstm_bwd(ptr(x), ptr(w_ih), ptr(w_hh), ptr(h0), ptr(c0), ptr(hs), ptr(cs), ptr(gates), ptr(d_hs),
                             ptr(d_hT), ptr(d_cT), T, B, I, H, int(reverse), ptr(dx), ptr(dw_ih), ptr(dw_hh), ptr(db),
                             ptr(dh0), ptr(dc0), ctx.mode, ptr(ws), ws.numel(), stream_ptr()), "gtg_lstm_bwd")
        return (dx, None if s_ih is not None else dw_ih, None if s_hh is not None else dw_hh, db, db, dh0, dc0, None,
                None)


def lstm(x, w_ih, w_hh, b_ih, b_hh, h0=None, c0=None, reverse=False, matrix_mode=None):
    """(hs (T, B, H), h_n (B, H), c_n (B, H)) of one LSTM direction; see LSTMFn."""
    return LSTMFn.apply(x, w_ih, w_hh, b_ih, b_hh, h0, c0, reverse, matrix_mode)


def proj_ce(H, W, b, targets, padidx, matrix_mode=None):
    """Per-token cross-entropy of the projected decoder outputs and the arg-max token; see ProjCEFn."""
    return ProjCEFn.apply(H, W, b, targets, padidx, matrix_mode)


# ---------------------------------------------------------------------------
# text branch: word drop, embedding lookup + dropout, dropout
# ---------------------------------------------------------------------------
def _rng_key():
    """(seed, stream id) for one randomised forward call: the seed is PyTorch's (``torch.manual_seed``), the stream
    id is drawn from PyTorch's default CPU generator -- no device work, no synchronisation, and a run repeats
    under the same ``torch.manual_seed`` exactly as stock dropout does.  The library's keep decisions are a pure
    function of (seed, stream id, element index), so the backward regenerates the forward's mask from this pair."""
    seed = torch.initial_seed() & 0xFFFFFFFFFFFFFFFF
    sid = int(torch.randint(0, 2 ** 62, (1,)).item())
    return seed, sid


@_on_tensor_device
def word_drop(x: torch.Tensor, padidx: int, p: float, keep: Optional[torch.Tensor] = None) -> torch.Tensor:
    """noise.word_drop (noise.py:12-22) on the device: x (T, B) int64 token ids; every token of a column but the
    first is dropped with probability p, the kept ones move up in order, the tail is filled with padidx.
    ``keep`` (T, B) bool / uint8: an explicit keep mask instead of the random one."""
    _lib.require_cuda(x)
    T, B = x.shape
    x = x.contiguous().long()
    out = torch.empty_like(x)
    k = None if keep is None else keep.to(device=x.device, dtype=torch.uint8).contiguous()
    seed, sid = (0, 0) if k is not None else _rng_key()
    check(_lib.lib().gtg_word_drop(ptr(x), T, B, int(padidx), float(p), seed, sid, ptr(k), ptr(out), stream_ptr()),
          "gtg_word_drop")
    return out


class EmbedDropoutFn(torch.autograd.Function):
    """drop(weight[ids]) (+ add broadcast over the leading dimension): text_vae.py:56 and :62 in one kernel, the
    backward an ordered scatter-add.  ids (T, B) int64, weight (V, D), add (B, D) or None -> (T, B, D).

    A weight whose ``grad_inplace`` attribute is true and whose ``.grad`` is already a dense tensor (JointTrainer's
    flat gradient buffer) receives its gradient by accumulation into that tensor -- the backward then returns None
    for it, and no (V, D) temporary is made; otherwise a dense (V, D) gradient is returned to autograd."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, ids, weight, add, p):
        L = _lib.lib()
        _lib.require_cuda(ids, weight, add)
        T, B = ids.shape
        V, D = weight.shape
        if D % 4:
            raise _lib.GtgError("embed_dropout: the embedding width must be a multiple of 4")
        w = _f32c(weight)
        a = _f32c(add)
        if a is not None and tuple(a.shape) != (B, D):
            raise _lib.GtgError("embed_dropout: add must be (B, D)")
        ids_c = ids.contiguous().long()
        out = torch.empty((T, B, D), dtype=torch.float32, device=w.device)
        err = torch.zeros(1, dtype=torch.int32, device=w.device) if EmbedDropoutFn.check_ids else None
        seed, sid = _rng_key() if p > 0 else (0, 0)
        check(L.gtg_embed_dropout_fwd(ptr(w), ptr(ids_c), T * B, V, D, float(p), seed, sid, ptr(a), B if a is not None else 0,
                                      ptr(out), ptr(err), stream_ptr()), "gtg_embed_dropout_fwd")
        if err is not None and int(err.item()):
            raise _lib.GtgError("embed_dropout: a token id lies outside the embedding table")
        ctx.save_for_backward(ids_c)
        ctx.weight = weight
        ctx.key = (float(p), seed, sid, V, D, B, a is not None)
        return out

    check_ids = False   # reading the flag back is a host sync: for tests and debugging

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dout):
        L = _lib.lib()
        (ids,) = ctx.saved_tensors
        p, seed, sid, V, D, B, has_add = ctx.key
        dout = _f32c(dout)
        R = ids.numel()
        weight = ctx.weight
        need_w, need_add = ctx.needs_input_grad[1], has_add and ctx.needs_input_grad[2]
        dw_ret = None
        target = None
        if need_w:
            g = weight.grad
            if getattr(weight, "grad_inplace", False) and g is not None and g.is_cuda and g.dtype == torch.float32 \
                    and g.is_contiguous() and g.shape == weight.shape:
                target = g                      # accumulated in place; autograd gets no gradient for the weight
            else:
                target = dw_ret = torch.zeros((V, D), dtype=torch.float32, device=dout.device)
        d_add = torch.empty((B, D), dtype=torch.float32, device=dout.device) if need_add else None
        if target is not None or d_add is not None:
            ws = _ws(L.gtg_embed_dropout_bwd_workspace_bytes(R, V, D, B if need_add else 0), dout.device)
            check(L.gtg_embed_dropout_bwd(ptr(dout), ptr(ids), R, V, D, p, seed, sid, ptr(target), ptr(d_add),
                                          B if need_add else 0, ptr(ws), ws.numel(), stream_ptr()), "gtg_embed_dropout_bwd")
        return None, dw_ret, d_add, None


def embed_dropout(ids, weight, p: float = 0.0, add=None):
    """(T, B, D) = dropout_p(weight[ids]) + add[None]; see EmbedDropoutFn."""
    return EmbedDropoutFn.apply(ids, weight, add, float(p))


class DropoutFn(torch.autograd.Function):
    """nn.Dropout(p) in training mode (text_vae.py:64): y = x * keep / (1 - p), the mask regenerated in backward."""

    @staticmethod
    @_on_tensor_device
    def forward(ctx, x, p):
        x = _f32c(x)
        if x.numel() % 4:
            raise _lib.GtgError("dropout: the element count must be a multiple of 4")
        y = torch.empty_like(x)
        seed, sid = _rng_key()
        check(_lib.lib().gtg_dropout(ptr(x), x.numel(), float(p), seed, sid, ptr(y), stream_ptr()), "gtg_dropout")
        ctx.key = (float(p), seed, sid)
        return y

    @staticmethod
    @_on_tensor_device
    def backward(ctx, dy):
        p, seed, sid = ctx.key
        dy = _f32c(dy)
        dx = torch.empty_like(dy)
        check(_lib.lib().gtg_dropout(ptr(dy), dy.numel(), p, seed, sid, ptr(dx), stream_ptr()), "gtg_dropout")
        return dx, None


def dropout(x, p: float, training: bool = True):
    """x when not training or p == 0, else DropoutFn."""
    if not training or p <= 0.0:
        return x
    return DropoutFn.apply(x, float(p))
