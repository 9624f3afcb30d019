"""Maximum-spanning-tree read-out on the GPU.

Mirrors /root/reference/src/maxspantree.py: ``mst_graph(graph)`` keeps the
reference's signature and return type (maxspantree.py:118); the work is done
by one wavefront per graph in csrc/mst.hip.  ``mst_graph_batched`` is the
batched form the evaluation path uses: it takes the decoder's output rows
directly, so GraphVAE.recover_adj_lower / recover_full_adj_from_lower
(/root/reference/src/graph_vae.py:61-69) never run on the host.
"""
from typing import List, Tuple

import torch

from . import _lib


def mst_graph_batched(x: torch.Tensor, nmax: int = None) -> Tuple[torch.Tensor, torch.Tensor]:
    """x: (G, N, N) matrices or (G, N(N+1)/2) upper-triangular rows, fp32 on the GPU.

    Returns ``(edges, count)``: edges (G, N-1, 2) int32 ordered (parent, vertex)
    pairs padded with -1, count (G,) int32.  No host synchronisation.
    """
    _lib.require_cuda(x)
    if x.dtype != torch.float32:
        raise _lib.GtgError("mst_graph_batched expects float32 weights (the reference decodes float32)")
    x = x.contiguous()
    L = _lib.lib()
    if x.dim() == 3:
        G, n, n2 = x.shape
        if n != n2:
            raise _lib.GtgError("square matrices expected")
        fn = L.gtg_mst_batched_full
    elif x.dim() == 2:
        G, m = x.shape
        if nmax is None:
            nmax = int(((8 * m + 1) ** 0.5 - 1) / 2)
        n = nmax
        if n * (n + 1) // 2 != m:
            raise _lib.GtgError(f"row length {m} is not n(n+1)/2 for n={n}")
        fn = L.gtg_mst_batched_triu
    else:
        raise _lib.GtgError("expected (G,N,N) or (G,M)")
    edges = torch.empty((G, max(n - 1, 0), 2), dtype=torch.int32, device=x.device)
    count = torch.empty((G,), dtype=torch.int32, device=x.device)
    if G > 0:
        with torch.cuda.device(x.device):   # launch on x's GPU even if another one is current
            _lib.check(fn(_lib.ptr(x), G, n, _lib.ptr(edges), _lib.ptr(count), _lib.stream_ptr()), "mst_graph_batched")
    return edges, count


def mst_graph(graph: torch.Tensor) -> List[Tuple[int, int]]:
    """Drop-in for the reference's ``mst_graph``: one (N, N) tensor in, the
    ordered Python list of (parent, vertex) pairs out."""
    if graph.dim() != 2 or graph.shape[0] != graph.shape[1]:
        raise _lib.GtgError("mst_graph expects one square (N, N) matrix")
    g = graph.detach().to(device="cuda", dtype=torch.float32)
    edges, count = mst_graph_batched(g.unsqueeze(0))
    k = int(count[0].item())
    return [(int(a), int(b)) for a, b in edges[0, :k].tolist()]
