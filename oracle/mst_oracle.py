"""CPU oracle for the maximum-spanning-tree read-out.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the product path (gtg_vae_amd) never does.

Restates the algorithm of the reference's ``mst_graph``
(/root/reference/src/maxspantree.py:118-163) on numpy float32 values:

* ``present_nodes``   <- helper_mst's diagonal scan        (maxspantree.py:129-134)
* ``prim_parents``    <- maximumSpanningTree/findMaxVertex (maxspantree.py:61-115, 7-31)
* ``mst_edges``       <- helper_mst + printMaximumSpanningTree (maxspantree.py:136-146, 35-58)
* ``full_from_triu``  <- GraphVAE.recover_adj_lower / recover_full_adj_from_lower
                         (/root/reference/src/graph_vae.py:61-69)

Parity pin: tests/test_oracle_mst.py checks this file against
tests/golden/mst_golden.json, which tests/golden/make_golden.py produced by
importing and running the reference's maxspantree.py in the build container.

The reference keeps ``weights`` as a mixed list of Python ints (+-sys.maxsize)
and 0-dim float32 tensors.  This restatement keeps that mix literally (ints vs
np.float32) so that the float-only sentinel used by the C oracle and the HIP
kernel is itself checked against a literal model of the original comparisons.
"""
from __future__ import annotations

import sys
from typing import List, Sequence, Tuple

import numpy as np

_MAXSIZE = sys.maxsize


def _gt(a, b) -> bool:
    """``a > b`` as Python/torch evaluate it when either side may be an int.

    int vs int stays an exact integer compare; as soon as a float32 tensor is
    involved torch converts the Python int to float32 (-2**63 for -maxsize)
    and compares in float32 (maxspantree.py:24,107).
    """
    if isinstance(a, int) and isinstance(b, int):
        return a > b
    with np.errstate(invalid="ignore"):
        return bool(np.float32(a) > np.float32(b))


def present_nodes(G: np.ndarray) -> List[int]:
    """Indices i with G[i,i] > 0.5, ascending (maxspantree.py:129-132)."""
    n = G.shape[0]
    return [i for i in range(n) if bool(G[i, i] > np.float32(0.5))]


def prim_parents(T: np.ndarray) -> List[int]:
    """Prim maximum spanning tree on a V x V float32 matrix, rooted at 0.

    Returns ``parent`` with parent[0] == -1.  A parent of -1 on any other
    vertex means "the last vertex", exactly as Python's negative indexing
    makes it behave in the reference (maxspantree.py:95,103,144).
    """
    V = T.shape[0]
    visited = [False] * V
    weights: list = [-_MAXSIZE] * V
    parent = [0] * V
    weights[0] = _MAXSIZE
    parent[0] = -1
    for _ in range(V - 1):
        # findMaxVertex: first index of the strictly greatest unvisited weight,
        # -1 when nothing beats -maxsize (maxspantree.py:11-31).
        idx, best = -1, -_MAXSIZE
        for i in range(V):
            if not visited[i] and _gt(weights[i], best):
                best, idx = weights[i], i
        visited[idx] = True  # idx == -1 marks the last vertex
        col = idx  # numpy negative indexing == torch negative indexing
        for j in range(V):
            t = T[j, col]
            if bool(t != np.float32(0)) and not visited[j]:
                if _gt(t, weights[j]):
                    weights[j] = np.float32(t)
                    parent[j] = idx
    return parent


def mst_edges(G: np.ndarray) -> List[Tuple[int, int]]:
    """Ordered edge list of the reference's ``mst_graph`` for one (N,N) matrix."""
    G = np.asarray(G, dtype=np.float32)
    P = present_nodes(G)
    V = len(P)
    if V == 0:
        return []
    T = G[np.ix_(P, P)]
    parent = prim_parents(T)
    return [(int(P[parent[i]]), int(P[i])) for i in range(1, V)]


def triu_index(i: int, j: int, n: int) -> int:
    """Row-major position of (i,j), i<=j, in the upper-triangular vector
    (the order ``adj[torch.triu(ones)==1] = l`` fills, graph_vae.py:64)."""
    return i * n - (i * (i - 1)) // 2 + (j - i)


def full_from_triu(row: Sequence[float], n: int) -> np.ndarray:
    """Symmetric (n,n) float32 matrix from an M=n(n+1)/2 vector
    (graph_vae.py:61-69: U + U^T - diag(U), evaluated in float32)."""
    U = np.zeros((n, n), dtype=np.float32)
    U[np.triu_indices(n)] = np.asarray(row, dtype=np.float32)
    with np.errstate(invalid="ignore", over="ignore"):
        return (U + U.T) - np.diag(np.diag(U))


def mst_edges_from_triu(row: Sequence[float], n: int) -> List[Tuple[int, int]]:
    return mst_edges(full_from_triu(row, n))


def edge_precision_recall(pred: Sequence[Tuple[int, int]], gold) -> Tuple[float, float]:
    """(recall, precision) of GraphVAE.graph_statistics (graph_vae.py:75-92).

    Returns (1, 1) when either list is empty, as the reference does.
    """
    gold = set(tuple(e) for e in gold)
    hits = sum(1 for (a, b) in pred if (a, b) in gold or (b, a) in gold)
    if len(gold) == 0 or len(pred) == 0:
        return 1.0, 1.0
    return hits / len(gold), hits / len(pred)


def pack_edges(edges: Sequence[Tuple[int, int]], n: int) -> Tuple[np.ndarray, int]:
    """(n-1, 2) int32 array padded with -1, and the edge count: the layout the
    batched C/HIP entry points write."""
    out = np.full((max(n - 1, 0), 2), -1, dtype=np.int32)
    for k, (a, b) in enumerate(edges):
        out[k, 0], out[k, 1] = a, b
    return out, len(edges)
