"""Generate tests/golden/noise_golden.npz by RUNNING the reference's word_drop (src/noise.py:12-22 imports with
torch + numpy alone).

    python tests/golden/make_noise_golden.py      # build container only

word_drop draws its keep flags from numpy's global generator, one `np.random.rand(T) > p` per sentence in
sentence order.  For every case the generator is seeded, the reference run, and then re-seeded to draw the same
flags again in the same order, so the fixture holds (tokens, keep flags, the reference's output): the oracle and
the HIP kernel take the flags as an input and must reproduce the output exactly.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from make_text_golden import make_tokens  # noqa: E402

CASES = {"small": (12, 7, 50, 0, 0.3, 1), "batch": (40, 33, 30522, 0, 0.3, 2), "long": (70, 5, 97, 0, 0.5, 3),
         "keep_all": (9, 4, 50, 0, 0.0, 4), "drop_most": (20, 6, 50, 3, 0.95, 5)}


def main():
    sys.path.insert(0, REF)
    import noise  # noqa

    out = {}
    for tag, (T, B, vocab, pad, p, seed) in CASES.items():
        x = torch.from_numpy(make_tokens(T, B, vocab, pad, seed=100 + seed))
        np.random.seed(seed)
        y = noise.noisy(pad, x, p, 0, 0, 0) if p > 0 else noise.word_drop(pad, x, p)
        np.random.seed(seed)
        keep = np.stack([np.random.rand(T) > p for _ in range(B)], axis=1)   # (T, B), the draws of noise.py:16
        out[tag + "__x"] = x.numpy()
        out[tag + "__keep"] = keep
        out[tag + "__out"] = y.numpy()
        out[tag + "__meta"] = np.array([pad, int(round(p * 1000))], dtype=np.int64)
    np.savez_compressed(os.path.join(HERE, "noise_golden.npz"), **out)
    print("wrote", len(out), "arrays")


if __name__ == "__main__":
    main()
