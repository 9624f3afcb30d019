"""Streams this package owns, and the lock that keeps a loader thread out of a graph capture.

``torch.cuda.Stream()`` does not create a stream: it hands out the next of 32 pooled HIP streams per
device and priority, round-robin (c10/hip/HIPStream.h), and ProcessGroupNCCL draws its collective
streams from the same pool.  Two draws 32 apart are therefore the SAME ``hipStream_t``.  A training
loop that draws a stream per epoch (a copy stream for its prefetcher) and one per capture sooner or
later has a worker thread issuing copies on the very handle a HIP-graph capture is open on: the
capture then ends with ``hipErrorStreamCaptureUnjoined``, or -- worse -- swallows the foreign copy
into the step's graph and replays it from freed memory on every step.

Hence two rules, both enforced here:

* every stream the package uses beside the caller's current one is created ONCE per (device, role)
  with ``hipStreamCreateWithFlags`` through the library (``gtg_stream_create``) and wrapped in a
  ``torch.cuda.ExternalStream``: its handle is in nobody's pool, so neither a later
  ``torch.cuda.Stream()`` nor an RCCL stream can be it.  Roles: ``capture`` (warm-up + HIP-graph
  capture of a trainer's step), ``copy`` (host-to-device copies of the prefetching loader),
  ``lstm`` (the reverse direction of the text encoder);
* ``device_work_lock`` is held by a capture from its warm-up to its end and by a loader thread
  around each pin -> copy -> record, so no other thread of the package issues device work while a
  capture is open (capture itself runs in ``thread_local`` error mode, which tolerates the threads
  the package does not own, such as the collectives' watchdog).

The reference has no counterpart: it is single-threaded and issues everything on PyTorch's current
stream (src/train.py:302-314).
"""
import ctypes
import threading
from typing import Dict, Tuple

import torch

from . import _lib

ROLES = ("capture", "copy", "lstm")

# Re-entrant: a capture that (indirectly) asks for another owned stream takes it again on the same thread.
device_work_lock = threading.RLock()

_streams: Dict[Tuple[int, str], "torch.cuda.ExternalStream"] = {}
_table_lock = threading.Lock()


def _device_index(device) -> int:
    dev = torch.device(device)
    if dev.type != "cuda":
        raise _lib.GtgError("owned streams exist on GPU devices only (got %s)" % dev)
    return torch.cuda.current_device() if dev.index is None else dev.index


def owned_stream(device, role: str) -> "torch.cuda.ExternalStream":
    """The package's stream for ``role`` on ``device``: created on first use, kept until the process ends,
    the same object on every call.  Its handle differs from every other owned stream's and from the
    device's default stream by construction; that is asserted, since everything above rests on it."""
    if role not in ROLES:
        raise _lib.GtgError("unknown stream role %r (one of %s)" % (role, ROLES))
    idx = _device_index(device)
    with _table_lock:
        s = _streams.get((idx, role))
        if s is None:
            raw = ctypes.c_void_p()
            _lib.check(_lib.lib().gtg_stream_create(idx, ctypes.byref(raw)), "gtg_stream_create")
            handle = int(raw.value or 0)
            taken = {int(o.cuda_stream) for (i, _r), o in _streams.items() if i == idx}
            taken.add(int(torch.cuda.default_stream(idx).cuda_stream))
            if handle == 0 or handle in taken:
                raise _lib.GtgError("gtg_stream_create returned a handle already in use (%#x)" % handle)
            s = torch.cuda.ExternalStream(handle, device=torch.device("cuda", idx))
            _streams[(idx, role)] = s
        return s


def owned_handles(device) -> Dict[str, int]:
    """{role: raw hipStream_t} of the streams created so far on ``device`` (for tests and diagnostics)."""
    idx = _device_index(device)
    with _table_lock:
        return {r: int(s.cuda_stream) for (i, r), s in _streams.items() if i == idx}
