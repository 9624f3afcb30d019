#!/usr/bin/env python3
"""Time the batched MST read-out on 65 536 graphs, N=64 (BASELINE configs[3]): mst_bench.py [reps] [library]."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

if len(sys.argv) > 2:          # an experimental build of the library (A/B runs)
    from gtg_vae_amd import _lib
    _lib.LIB_PATH = os.path.abspath(sys.argv[2])

print(json.dumps(bench.mst_bench(torch.device("cuda"), reps=int(sys.argv[1]) if len(sys.argv) > 1 else 10)))
