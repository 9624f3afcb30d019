"""Host-side batch builder: the input contract of the hot path.

Mirrors ``loaddata.prep_data_list`` (/root/reference/src/train.py:174-213) and
the torch_geometric ``DataLoader`` collate used at train.py:549 (node features
concatenated, edge indices offset by graph, a ``batch`` vector of graph ids),
without depending on torch_geometric.  Also generates the synthetic AMR-shaped
graphs used by the benchmark and the tests.
"""
import json
from dataclasses import dataclass, field
from typing import Iterable, List, Optional, Sequence, Set, Tuple

import numpy as np
import torch


@dataclass
class GraphData:
    """One graph, as train.py:203-210 builds it (x is the identity and is not stored)."""
    edge_index: torch.Tensor          # (2, E_g) int64
    gold_edges: Set[Tuple[int, int]]
    adj_padded: np.ndarray            # (maxnode, maxnode) float64
    snts: str = ""
    num_nodes: int = 0
    maxnode: int = 0
    tokens: Optional[List[int]] = None  # token ids of ``snts`` (what train.py:305 gets from the tokenizer), if known

    @property
    def x(self) -> torch.Tensor:
        return torch.eye(self.maxnode, dtype=torch.float)

    def adj32(self) -> torch.Tensor:
        """adj_padded as a float32 tensor (what the model consumes), converted once."""
        if getattr(self, "_adj32", None) is None:
            self._adj32 = torch.from_numpy(np.asarray(self.adj_padded, dtype=np.float32))
        return self._adj32

    def gold_row(self) -> torch.Tensor:
        """(maxnode,) int64 bit masks of the gold edges, built once."""
        if getattr(self, "_gold_row", None) is None:
            self._gold_row = gold_bitmask([self.gold_edges], self.maxnode)[0]
        return self._gold_row


def prep_graph(graph: dict, maxnode: int) -> Optional[GraphData]:
    """train.py:177-211 for one MRP graph dict; None when the graph is skipped."""
    num_of_nodes = len(graph["nodes"])
    if num_of_nodes > maxnode or "edges" not in graph:
        return None
    edges: List[Tuple[int, int]] = []
    seen = set()
    gold = []
    adj = np.zeros((num_of_nodes, num_of_nodes))
    for edge in graph["edges"]:
        s, t = edge["source"], edge["target"]
        # both directions are appended every time (duplicates kept, train.py:187-188);
        # a self loop is appended only if that pair is not in the list yet (train.py:189-192)
        for pair in ((s, t), (t, s)):
            edges.append(pair)
            seen.add(pair)
        for v in (s, t):
            if (v, v) not in seen:
                edges.append((v, v))
                seen.add((v, v))
        if s != t:
            gold.append((s, t))
            adj[s, t] = 1
            adj[t, s] = 1
    adj = adj + np.identity(num_of_nodes)
    adj_padded = np.zeros((maxnode, maxnode))
    adj_padded[:num_of_nodes, :num_of_nodes] = adj
    ei = torch.tensor(edges, dtype=torch.long).reshape(-1, 2).t().contiguous()
    return GraphData(ei, set(gold), adj_padded, graph.get("input", ""), num_of_nodes, maxnode, graph.get("tokens"))


def prep_data_list(graphs: Iterable[dict], maxnode: int) -> List[GraphData]:
    out = []
    for g in graphs:
        d = prep_graph(g, maxnode)
        if d is not None:
            out.append(d)
    return out


def load_mrp(path: str) -> List[dict]:
    """One JSON object per line (train.py:218-221)."""
    with open(path, "r") as f:
        return [json.loads(l) for l in f if l.strip()]


@dataclass
class GraphBatch:
    """A collated batch.  ``x`` is None: node features are stacked identities
    (train.py:203-204), which the encoder exploits; ``dense_x()`` materialises
    them for callers that want the reference's tensor."""
    edge_index: torch.Tensor            # (2, E) int64, node id = graph * nmax + local id
    batch: torch.Tensor                 # (B * nmax,) int64 graph id per node
    adj_padded: torch.Tensor            # (B, nmax, nmax) float32 reconstruction targets
    gold_edges: List[Set[Tuple[int, int]]]
    snts: List[str]
    num_graphs: int
    nmax: int
    num_nodes: torch.Tensor = None      # (B,) int32
    x: Optional[torch.Tensor] = None
    gold_bits: Optional[torch.Tensor] = field(default=None)  # (B, nmax) int64 bit masks
    global_graphs: Optional[int] = None  # size of the global batch this one is a rank's shard of (None: not a shard)
    dec_seq: Optional[torch.Tensor] = None  # (B, T) int64 token ids padded with 0 to the batch's longest sentence

    def dense_x(self) -> torch.Tensor:
        return torch.eye(self.nmax, dtype=torch.float, device=self.edge_index.device).repeat(self.num_graphs, 1)

    def to(self, device) -> "GraphBatch":
        mv = lambda t: None if t is None else t.to(device, non_blocking=True)
        return GraphBatch(mv(self.edge_index), mv(self.batch), mv(self.adj_padded), self.gold_edges, self.snts,
                          self.num_graphs, self.nmax, mv(self.num_nodes), mv(self.x), mv(self.gold_bits),
                          self.global_graphs, mv(self.dec_seq))

    def pin_memory(self) -> "GraphBatch":
        """Host tensors in page-locked memory, so that ``to(device)`` is an asynchronous DMA."""
        pin = lambda t: None if t is None else (t if t.is_cuda or t.is_pinned() else t.pin_memory())
        return GraphBatch(pin(self.edge_index), pin(self.batch), pin(self.adj_padded), self.gold_edges, self.snts,
                          self.num_graphs, self.nmax, pin(self.num_nodes), pin(self.x), pin(self.gold_bits),
                          self.global_graphs, pin(self.dec_seq))

    def tensors(self):
        return [t for t in (self.edge_index, self.batch, self.adj_padded, self.num_nodes, self.x, self.gold_bits,
                            self.dec_seq) if t is not None]

    def __len__(self):
        return self.num_graphs


def gold_bitmask(gold_edges: Sequence[Set[Tuple[int, int]]], nmax: int) -> torch.Tensor:
    """(B, nmax) int64: bit b of word [g, a] is set iff (a, b) is a gold edge of graph g."""
    out = np.zeros((len(gold_edges), nmax), dtype=np.uint64)
    for g, es in enumerate(gold_edges):
        for a, b in es:
            out[g, a] |= np.uint64(1) << np.uint64(b)
    return torch.from_numpy(out.view(np.int64))


def collate(data_list: Sequence[GraphData], nmax: int) -> GraphBatch:
    B = len(data_list)
    eis = [d.edge_index + g * nmax for g, d in enumerate(data_list)]
    ei = torch.cat(eis, dim=1) if eis else torch.zeros((2, 0), dtype=torch.long)
    batch = torch.arange(B, dtype=torch.long).repeat_interleave(nmax)
    adj = torch.stack([d.adj32() for d in data_list]) if B else torch.zeros((0, nmax, nmax))
    gold = [d.gold_edges for d in data_list]
    nn = torch.tensor([d.num_nodes for d in data_list], dtype=torch.int32)
    bits = torch.stack([d.gold_row() for d in data_list]) if B else torch.zeros((0, nmax), dtype=torch.int64)
    return GraphBatch(ei.contiguous(), batch, adj, gold, [d.snts for d in data_list], B, nmax, nn, None, bits,
                      dec_seq=pad_tokens([d.tokens for d in data_list]))


def pad_tokens(token_lists, pad: int = 0) -> Optional[torch.Tensor]:
    """(B, T) int64, every sentence padded with ``pad`` to the longest of the batch -- what
    ``tokenizer(snts, padding=True)['input_ids']`` hands the model at train.py:305.  None when a
    graph carries no token ids."""
    if not token_lists or any(t is None for t in token_lists):
        return None
    T = max(len(t) for t in token_lists)
    out = np.full((len(token_lists), T), pad, dtype=np.int64)
    for b, t in enumerate(token_lists):
        out[b, :len(t)] = t
    return torch.from_numpy(out)


class PackedGraphs:
    """A list of GraphData packed once into a few flat host tensors, so that a batch is a handful of vectorised
    gathers instead of a Python loop over its graphs (which is what makes per-batch collation the bottleneck of a
    host-side loader: torch_geometric's collate, train.py:549, walks the batch's Data objects in Python too):
    every graph's edge list with LOCAL node ids, concatenated, with start offsets and the owning graph of every
    edge; the padded fp32 targets; gold-edge bit masks; node counts; token ids padded to the longest sentence
    with their lengths.  ``batch(ids)`` returns exactly what ``collate([data[i] for i in ids], nmax)`` returns."""

    def __init__(self, data_list: Sequence[GraphData], nmax: int):
        self.data = list(data_list)
        self.nmax, self.num_graphs = nmax, len(self.data)
        lens = np.array([d.edge_index.shape[1] for d in self.data], dtype=np.int64)
        start = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum(lens, out=start[1:])
        self.edge_len = torch.from_numpy(lens)
        self.edge_start = torch.from_numpy(start)
        self.edges_local = (torch.cat([d.edge_index for d in self.data], dim=1) if self.data
                            else torch.zeros((2, 0), dtype=torch.long)).contiguous()
        self.adj_all = (torch.stack([d.adj32() for d in self.data]) if self.data
                        else torch.zeros((0, nmax, nmax))).contiguous()
        self.gold_bits = gold_bitmask([d.gold_edges for d in self.data], nmax)
        self.num_nodes = torch.tensor([d.num_nodes for d in self.data], dtype=torch.int32)
        self.edge_graph = torch.repeat_interleave(torch.arange(self.num_graphs, dtype=torch.int64), self.edge_len)
        self.tokens = None
        if self.data and all(d.tokens is not None for d in self.data):
            self.tokens = pad_tokens([d.tokens for d in self.data])
            self.token_len = torch.tensor([len(d.tokens) for d in self.data], dtype=torch.int64)
        self.pinned = False

    def __len__(self):
        return self.num_graphs

    def pin(self):
        """Move the large arrays to page-locked memory, once: a batch of consecutive graphs (an unshuffled
        loader's) is then made of views of pinned memory, and its host-to-device copy is an asynchronous DMA
        with no per-batch pinning copy."""
        if not self.pinned and torch.cuda.is_available():
            self.adj_all, self.gold_bits, self.num_nodes = self.adj_all.pin_memory(), self.gold_bits.pin_memory(), \
                self.num_nodes.pin_memory()
            if self.tokens is not None:
                self.tokens = self.tokens.pin_memory()
            self.pinned = True
        return self

    def batch(self, ids) -> "GraphBatch":
        ids = torch.as_tensor(ids, dtype=torch.int64)
        B, n = int(ids.numel()), self.nmax
        if B == 0:
            return collate([], n)
        a = int(ids[0])
        if int(ids[-1]) - a == B - 1 and bool((ids == torch.arange(a, a + B)).all()):
            # consecutive graphs: slices (views) of the packed arrays, one vectorised renumbering of the edges
            b, s, e = a + B, int(self.edge_start[a]), int(self.edge_start[a + B])
            ei = self.edges_local[:, s:e] + (self.edge_graph[s:e] - a) * n
            dec_seq = None
            if self.tokens is not None:
                dec_seq = self.tokens[a:b, : int(self.token_len[a:b].max())].contiguous()
            return GraphBatch(ei, torch.arange(B, dtype=torch.long).repeat_interleave(n), self.adj_all[a:b],
                              [d.gold_edges for d in self.data[a:b]], [d.snts for d in self.data[a:b]], B, n,
                              self.num_nodes[a:b], None, self.gold_bits[a:b], dec_seq=dec_seq)
        lens = self.edge_len[ids]
        # position of every edge of the batch in the packed list: the graph's start + 0..len-1
        slot = torch.repeat_interleave(torch.arange(B, dtype=torch.int64), lens)          # batch slot of each edge
        first = torch.cumsum(lens, 0) - lens                                               # where each graph's run begins
        pos = self.edge_start[ids][slot] + (torch.arange(int(lens.sum()), dtype=torch.int64) - first[slot])
        ei = self.edges_local[:, pos] + slot * n
        idl = ids.tolist()
        dec_seq = None
        if self.tokens is not None:
            dec_seq = self.tokens[ids][:, : int(self.token_len[ids].max())].contiguous()
        return GraphBatch(ei.contiguous(), torch.arange(B, dtype=torch.long).repeat_interleave(n), self.adj_all[ids],
                          [self.data[i].gold_edges for i in idl], [self.data[i].snts for i in idl], B, n,
                          self.num_nodes[ids], None, self.gold_bits[ids], dec_seq=dec_seq)


class DataLoader:
    """Minimal stand-in for torch_geometric's DataLoader(dataset, batch_size) (train.py:549-550).  The dataset is
    packed once (PackedGraphs); every batch is then assembled by vectorised gathers."""

    def __init__(self, dataset: Sequence[GraphData], batch_size: int, nmax: int = None, shuffle: bool = False, seed: int = 0):
        self.dataset = list(dataset)
        self.batch_size = batch_size
        self.nmax = nmax if nmax is not None else (self.dataset[0].maxnode if self.dataset else 0)
        self.shuffle = shuffle
        self._rng = np.random.default_rng(seed)
        self._packed = None

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def packed(self) -> PackedGraphs:
        if self._packed is None:
            self._packed = PackedGraphs(self.dataset, self.nmax)
        return self._packed

    def __iter__(self):
        self.packed()
        order = np.arange(len(self.dataset))
        if self.shuffle:
            self._rng.shuffle(order)
        for i in range(0, len(order), self.batch_size):
            yield self._packed.batch(order[i:i + self.batch_size])


class DeviceDataset:
    """The whole dataset in HBM (an AMR corpus of 60 k graphs at N = 64 is about 1 GB of targets; the card has
    288 GB): every graph's edge list with local node ids, concatenated, with start offsets; the padded
    targets; the gold-edge bit masks.  A batch is then a list of graph ids, gathered on the device by
    ``GraphVAETrainer.load_indices`` -- no per-batch collate, pinning or PCIe copy."""

    def __init__(self, data_list: Sequence[GraphData], nmax: int, device="cuda"):
        self.nmax, self.num_graphs = nmax, len(data_list)
        dev = torch.device(device)
        lens = [d.edge_index.shape[1] for d in data_list]
        start = np.zeros(len(lens) + 1, dtype=np.int64)
        np.cumsum(lens, out=start[1:])
        ei = torch.cat([d.edge_index for d in data_list], dim=1) if data_list else torch.zeros((2, 0), dtype=torch.long)
        self.max_edges_per_graph = max(lens) if lens else 0
        self.total_edges = int(start[-1])
        self.edges_local = ei.contiguous().to(dev)
        self.edge_start = torch.from_numpy(start).to(dev)
        self.adj_all = torch.from_numpy(np.stack([d.adj_padded for d in data_list]).astype(np.float32)).to(dev)
        self.gold_bits = gold_bitmask([d.gold_edges for d in data_list], nmax).to(dev)
        self.gold_edges = [d.gold_edges for d in data_list]

    def __len__(self):
        return self.num_graphs

    def batches(self, batch_size: int, shuffle: bool = False, generator: torch.Generator = None):
        """Device tensors of graph ids, one per batch (sequential like the reference's loader, or a permutation)."""
        dev = self.adj_all.device
        order = torch.randperm(self.num_graphs, generator=generator).to(dev) if shuffle else \
            torch.arange(self.num_graphs, device=dev)
        for i in range(0, self.num_graphs, batch_size):
            yield order[i:i + batch_size]


class DeviceLoader:
    """Iterates a DeviceDataset as batches of graph ids (device int64 tensors); ``rank``/``world`` take this
    rank's contiguous share of every batch (sizes differ by at most one; on a ragged last batch a share can be
    empty) and tag it with the global batch size.  ``GraphVAETrainer.load_indices`` consumes the ids."""

    def __init__(self, dataset: "DeviceDataset", batch_size: int, shuffle: bool = False, seed: int = 0, rank: int = 0,
                 world: int = 1):
        self.dataset, self.batch_size, self.shuffle = dataset, batch_size, shuffle
        self.rank, self.world = rank, world
        self.sharded = True
        self._gen = torch.Generator().manual_seed(seed)

    def __len__(self):
        return (len(self.dataset) + self.batch_size - 1) // self.batch_size

    def __iter__(self):
        for ids in self.dataset.batches(self.batch_size, self.shuffle, self._gen):
            if self.world > 1:
                total = len(ids)
                base, rem = divmod(total, self.world)
                lo = self.rank * base + min(self.rank, rem)
                ids = ids[lo:lo + base + (1 if self.rank < rem else 0)]   # may be empty on the ragged last batch
                ids.global_graphs = total    # the trainer weights this shard's gradient by len(ids) * world / total
            yield ids


class DevicePrefetcher:
    """Iterates a loader of host GraphBatch objects and yields them on the GPU,
    ``depth`` batches ahead: a worker thread collates (the loader's own work),
    pins and starts the host-to-device copies on a side stream while the step
    of the previous batch runs; the consumer's stream only waits for the copy
    event of the batch it is handed.  The batches are the loader's, in order.

    The side stream is the package's own copy stream of the device (streams.owned_stream: one per device,
    created with hipStreamCreateWithFlags, so it can never be the pooled stream a HIP-graph capture or a
    collective runs on), and the worker issues its device work under ``streams.device_work_lock``, which a
    trainer holds while it captures its step: the worker waits out a capture instead of running into it."""

    def __init__(self, loader, device, depth: int = 2):
        dev = torch.device(device)
        if dev.type == "cuda" and dev.index is None:
            dev = torch.device("cuda", torch.cuda.current_device())
        self.loader, self.device, self.depth = loader, dev, max(1, depth)
        self.dataset = getattr(loader, "dataset", None)
        inner = getattr(loader, "loader", loader)      # through train.ShardedLoader
        if hasattr(inner, "packed"):
            inner.packed().pin()                       # batches of consecutive graphs become views of pinned memory

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        import queue
        import threading

        from .streams import device_work_lock, owned_stream

        q: "queue.Queue" = queue.Queue(self.depth)
        copy_stream = owned_stream(self.device, "copy")
        stop = threading.Event()

        def put(item):
            while not stop.is_set():
                try:
                    q.put(item, timeout=0.1)
                    return True
                except queue.Full:
                    continue
            return False

        def worker():
            try:
                torch.cuda.set_device(self.device)
                for host in self.loader:          # collate: host work, outside the lock
                    with device_work_lock:
                        pinned = host.pin_memory()
                        with torch.cuda.stream(copy_stream):
                            dev = pinned.to(self.device)
                            done = torch.cuda.Event()
                            done.record(copy_stream)
                    if not put((dev, done, pinned)):   # the pinned source stays alive until the copy is consumed
                        return
                put(None)
            except BaseException as e:  # surfaces in the consumer
                put(e)

        t = threading.Thread(target=worker, daemon=True)
        t.start()
        try:
            while True:
                item = q.get()
                if item is None:
                    break
                if isinstance(item, BaseException):
                    raise item
                dev, done, _pinned = item
                cur = torch.cuda.current_stream(self.device)
                cur.wait_event(done)
                for ten in dev.tensors():
                    ten.record_stream(cur)   # allocated under the copy stream, used on this one
                yield dev
        finally:
            stop.set()


# ---------------------------------------------------------------------------
# synthetic AMR-shaped graphs
# ---------------------------------------------------------------------------
def synth_mrp_graph(rng: np.random.Generator, nmax: int) -> dict:
    """n ~ U{max(2, nmax/4) .. nmax}; a random recursive tree plus floor(0.15 n)
    re-entrancy edges between random distinct nodes (duplicates allowed)."""
    lo = max(2, nmax // 4)
    n = int(rng.integers(lo, nmax + 1)) if nmax >= lo else nmax
    edges = [{"source": int(rng.integers(0, t)), "target": t} for t in range(1, n)]
    for _ in range(int(0.15 * n)):
        a, b = rng.choice(n, size=2, replace=False)
        edges.append({"source": int(a), "target": int(b)})
    return {"nodes": [{"id": i} for i in range(n)], "edges": edges, "input": "synthetic sentence %d" % n}


def synth_graph_list(num: int, nmax: int, seed: int = 1234, vocab: int = 0, max_len: int = 40) -> List[GraphData]:
    """``vocab`` > 0 also attaches a synthetic sentence to every graph (``synth_tokens``)."""
    rng = np.random.default_rng(seed)
    out = prep_data_list([synth_mrp_graph(rng, nmax) for _ in range(num)], nmax)
    if vocab:
        for d in out:
            d.tokens = synth_tokens(rng, vocab, max_len)
    return out


def synth_tokens(rng: np.random.Generator, vocab: int = 30522, max_len: int = 40) -> List[int]:
    """A sentence shaped like bert-base-uncased output: length T ~ U{8..max_len} including the start (101)
    and end (102) symbols, word-piece ids uniform in [1000, vocab); small vocabularies use [3, vocab)."""
    T = int(rng.integers(min(8, max_len), max_len + 1))
    lo = 1000 if vocab > 2000 else 3
    body = rng.integers(lo, vocab, size=max(T - 2, 0)).tolist()
    first, last = (101, 102) if vocab > 102 else (1, 2)
    return [first] + body + [last]


def synth_graph_batch(B: int, nmax: int, seed: int = 1234) -> GraphBatch:
    return collate(synth_graph_list(B, nmax, seed), nmax)
