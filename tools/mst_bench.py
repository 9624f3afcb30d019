#!/usr/bin/env python3
"""Time the batched MST read-out on 65 536 graphs, N=64 (BASELINE configs[3])."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench

print(json.dumps(bench.mst_bench(torch.device("cuda"), reps=int(sys.argv[1]) if len(sys.argv) > 1 else 10)))
