#!/usr/bin/env python3
"""Shows, on one GPU, why a stream drawn with torch.cuda.Stream() cannot be trusted to be a new one, and that
the package's own streams are outside that pool.  Prints one JSON line.

    python tools/stream_pool_probe.py
"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

import torch  # noqa: E402

from gtg_vae_amd.streams import ROLES, owned_handles, owned_stream  # noqa: E402


def main():
    torch.cuda.set_device(0)
    if torch.cuda.graph.default_capture_stream is None:
        torch.cuda.graph.default_capture_stream = torch.cuda.Stream()   # what torch.cuda.graph() does on first use
    capture_default = int(torch.cuda.graph.default_capture_stream.cuda_stream)
    draws = [int(torch.cuda.Stream().cuda_stream) for _ in range(66)]
    first_repeat = next(i for i in range(1, len(draws)) if draws[i] == draws[0])
    hits = [i for i, h in enumerate(draws) if h == capture_default]
    for r in ROLES:
        owned_stream("cuda", r)
    own = owned_handles("cuda")
    print(json.dumps({
        "pool_period_draws": first_repeat,
        "distinct_handles_in_66_draws": len(set(draws)),
        "torch_default_capture_stream": hex(capture_default),
        "draws_that_returned_the_default_capture_stream": hits,
        "owned": {k: hex(v) for k, v in own.items()},
        "owned_in_pool": sorted(k for k, v in own.items() if v in set(draws) or v == capture_default),
    }))


if __name__ == "__main__":
    main()
