"""Data-parallel plumbing: shard a batch by graph, all-reduce flat gradient
buckets (RCCL over xGMI on the GPU; gloo in the CPU tests).

The reference is single-process (no DDP/NCCL call exists in it); this layer
is new.  One process per GPU; every rank holds the full replicated parameter
and optimiser state; each step exchanges exactly the flat fp32 gradient
buffer, split in two buckets so that the first (decoder + latent head, whose
gradients are final before the encoder's backward starts) is reduced while
the encoder's backward GEMMs still run.
"""
import os
from typing import List, Optional, Sequence, Tuple

import torch
import torch.distributed as dist


def env_world() -> Tuple[int, int, int]:
    """(rank, local_rank, world_size) from the torchrun environment."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def local_device_index(local_rank: int) -> int:
    """GPU index for this rank: LOCAL_RANK, wrapped when fewer GPUs than ranks are visible."""
    n = torch.cuda.device_count()
    return local_rank % n if n > 0 else 0


# A rank that stops answering must end the job, not hang it: every collective (and the rendezvous) gives up
# after this many seconds, RCCL's watchdog then aborts the process with a non-zero exit code and the launcher
# (torch.distributed.run, no restarts) takes the other ranks down.  PyTorch's default is ten minutes.
DIST_TIMEOUT_S = 300.0


def init_process_group(backend: Optional[str] = None, timeout_s: Optional[float] = None) -> Tuple[int, int, int]:
    """Join the job the torchrun environment describes (a no-op for a single process).  With RCCL the
    communicator is created here, eagerly (``device_id``), so that its set-up -- allocations, stream and event
    creation -- is over before any step is captured into a HIP graph."""
    import datetime

    rank, local_rank, world = env_world()
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            # GTG_DIST_BACKEND=gloo lets several ranks share one GPU for rehearsals (RCCL wants one GPU per rank)
            backend = os.environ.get("GTG_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
        if timeout_s is None:
            timeout_s = float(os.environ.get("GTG_DIST_TIMEOUT_S") or DIST_TIMEOUT_S)
        timeout = datetime.timedelta(seconds=timeout_s)
        if backend == "nccl":
            if torch.cuda.device_count() < 1:
                raise RuntimeError("the nccl (RCCL) backend needs a GPU per rank; none is visible")
            index = local_device_index(local_rank)
            torch.cuda.set_device(index)
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout,
                                    device_id=torch.device("cuda", index))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world, timeout=timeout)
    return rank, local_rank, world


def shard_range(num_graphs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous shard [lo, hi) of a global batch; sizes differ by at most one."""
    base, rem = divmod(num_graphs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_weight(local_graphs: int, global_graphs: int, world: int) -> float:
    """Factor for a rank's mean-over-its-shard gradient so that sum all-reduce followed by 1/world gives the
    mean over the global batch: B_local * world / B_global (1 when the shards are equal, 0 for an empty one)."""
    return float(local_graphs) * world / float(global_graphs) if global_graphs else 1.0


class GradBuckets:
    """Views of one flat gradient buffer, reduced bucket by bucket.

    ``bounds`` are element offsets [0, b1, ..., numel].  ``reduce(i)`` launches
    the sum all-reduce of bucket i asynchronously; ``finish()`` waits for all of
    them.  The 1/world scaling is left to the optimiser kernel (it multiplies
    the gradient by ``scale`` on load), so no extra pass over the buffer runs.
    """

    def __init__(self, flat_grad: torch.Tensor, bounds: Sequence[int], group=None):
        assert bounds[0] == 0 and bounds[-1] == flat_grad.numel()
        self.flat = flat_grad
        self.views = [flat_grad[a:b] for a, b in zip(bounds[:-1], bounds[1:])]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self._pending: List = []

    @property
    def scale(self) -> float:
        return 1.0 / self.world

    def reduce(self, i: int) -> None:
        if self.world == 1:
            return
        v = self.views[i]
        if v.is_cuda and dist.get_backend(self.group) == "gloo":
            # rehearsal path (several ranks on one GPU): this build's gloo does not reduce GPU tensors
            host = v.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
            v.copy_(host)
            return
        self._pending.append(dist.all_reduce(v, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def finish(self) -> None:
        for w in self._pending:
            w.wait()
        self._pending.clear()

    def reduce_all(self) -> None:
        for i in range(len(self.views)):
            self.reduce(i)
        self.finish()


def all_reduce_small(values: torch.Tensor, op=None, group=None) -> torch.Tensor:
    """All-reduce of a small reporting tensor; returns it on the CPU.  Goes
    through host memory when the backend is gloo."""
    if not (dist.is_initialized() and dist.get_world_size(group) > 1):
        return values.cpu()
    op = dist.ReduceOp.SUM if op is None else op
    if dist.get_backend(group) == "gloo":
        host = values.cpu()
        dist.all_reduce(host, op=op, group=group)
        return host
    dist.all_reduce(values, op=op, group=group)
    return values.cpu()
