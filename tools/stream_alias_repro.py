#!/usr/bin/env python3
"""Reproduces, deterministically, the capture failure that pooled streams cause -- by putting the OLD stream
policy back for one run: the capture on PyTorch's default capture stream, the prefetcher's copy stream a fresh
torch.cuda.Stream() per epoch, no lock between the loader thread and the capture.  The pool pointer is placed
so that the prefetcher's draw IS the capture stream.  Then the same loop with the package's real policy.

    python tools/stream_alias_repro.py          # prints one JSON line: {"old_policy": ..., "new_policy": ...}
"""
import contextlib
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from gtg_vae_amd import streams, trainer  # noqa: E402
from gtg_vae_amd.data import DataLoader, DevicePrefetcher, synth_graph_list  # noqa: E402
from gtg_vae_amd.train import train  # noqa: E402


def place_pool_pointer():
    if torch.cuda.graph.default_capture_stream is None:
        torch.cuda.graph.default_capture_stream = torch.cuda.Stream()
    target = int(torch.cuda.graph.default_capture_stream.cuda_stream)
    drawn = [int(torch.cuda.Stream().cuda_stream) for _ in range(32)]
    for _ in range(drawn.index(target)):
        torch.cuda.Stream()
    return target


COPY_STREAM = [None]    # old policy: the pooled stream the prefetcher drew last


class GatedLoader:
    """Holds the second batch back until a capture is open, and reports when the worker has come back for the
    third (i.e. has issued the second batch's copies): the loader thread's device work is thereby placed INSIDE
    the capture window instead of wherever the scheduler happens to put it."""

    def __init__(self, loader):
        import threading

        self.loader, self.dataset = loader, loader.dataset
        self.gate, self.issued = threading.Event(), threading.Event()

    def __len__(self):
        return len(self.loader)

    def __iter__(self):
        for i, b in enumerate(self.loader):
            if i == 1:
                self.gate.wait(5.0)
            if i == 2:
                # the second batch's copies have been issued: was the stream they went to under capture then?
                from gtg_vae_amd import streams as st

                with torch.cuda.stream(st.owned_stream("cuda", "copy") if COPY_STREAM[0] is None else COPY_STREAM[0]):
                    self.copy_stream_was_capturing = bool(torch.cuda.is_current_stream_capturing())
                self.issued.set()
            yield b


def run(loader, N):
    tr = trainer.GraphVAETrainer(N, 16, 32, N, device="cuda", seed=1)
    loader = GatedLoader(loader)
    segments = tr._segments

    def gated_segments(B, update, backward=True):
        segs = segments(B, update, backward)

        def first(st, f=segs[0]):
            if torch.cuda.is_current_stream_capturing():
                loader.gate.set()                 # let the loader thread go ...
                loader.issued.wait(1.0)           # ... and give it a second to issue its copies (it cannot, under the lock)
            return f(st)
        return [first] + segs[1:]

    tr._segments = gated_segments
    try:
        hist = train(DevicePrefetcher(loader, "cuda", depth=3), None, tr, epochs=4,
                     generator=torch.Generator().manual_seed(0), quiet=True, use_graph=True)
        torch.cuda.synchronize()
        return {"ok": True, "loss": hist[-1]["loss"], "captures": tr.captures,
                "loader_copied_into_a_capturing_stream": loader.copy_stream_was_capturing}
    except Exception as e:  # noqa: BLE001
        return {"ok": False, "error": "%s: %s" % (type(e).__name__, str(e).splitlines()[0][:200])}


def main():
    torch.cuda.set_device(0)
    N = 12
    loader = DataLoader(synth_graph_list(150, N, seed=3), batch_size=32, nmax=N)
    out = {}
    place_pool_pointer()
    out["new_policy"] = run(loader, N)

    # ---- the old policy, for this run only
    real = (streams.owned_stream, trainer.owned_stream, streams.device_work_lock, trainer.device_work_lock)

    def pooled(device, role):
        if role == "capture":
            return torch.cuda.graph.default_capture_stream
        COPY_STREAM[0] = torch.cuda.Stream()
        return COPY_STREAM[0]

    streams.owned_stream = trainer.owned_stream = pooled
    streams.device_work_lock = trainer.device_work_lock = contextlib.nullcontext()
    target = place_pool_pointer()
    out["old_policy"] = dict(run(loader, N), capture_stream=hex(target), copy_stream=hex(int(COPY_STREAM[0].cuda_stream)))
    streams.owned_stream, trainer.owned_stream, streams.device_work_lock, trainer.device_work_lock = real
    print(json.dumps(out))


if __name__ == "__main__":
    main()
