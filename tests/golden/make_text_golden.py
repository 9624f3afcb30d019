"""Generate tests/golden/textvae_golden.npz by RUNNING the reference's text
branch (src/text_vae.py, src/noise.py import with torch + numpy alone).

    python tests/golden/make_text_golden.py      # build container only

The DAE has fixed 512/1024-wide layers (25 M parameters), too large to store,
so the weights are not a fixture: every parameter is filled from an integer
hash of (parameter name order, element index) -- `fill_parameters` below, which
tests/test_text_branch.py imports -- and only inputs and outputs are stored.
Stored: token ids, the encoder output, decoder logits and final LSTM state,
loss_rec, accuracy (eval mode: no dropout, no word noise), and the
reference's loss_kl / log_prob on fixed inputs.
"""
import os
import sys

import numpy as np
import torch

REF = "/root/reference/src"
HERE = os.path.dirname(os.path.abspath(__file__))


def fill_parameters(module, scale=0.08):
    """Deterministic weights: value = scale * (hash(k, i) / 2^32 - 0.5) * 2 for
    element i of the k-th state_dict entry (sorted by name).  Pure integer
    arithmetic, identical on every platform."""
    sd = module.state_dict()
    with torch.no_grad():
        for k, name in enumerate(sorted(sd.keys())):
            t = sd[name]
            i = np.arange(t.numel(), dtype=np.uint64)
            h = (i * np.uint64(2654435761) + np.uint64(k + 1) * np.uint64(40503) + (i >> np.uint64(7)) * np.uint64(97)) % np.uint64(1 << 32)
            v = (h.astype(np.float64) / float(1 << 32) - 0.5) * 2.0 * scale
            t.copy_(torch.from_numpy(v.astype(np.float32)).view(t.shape))


def make_tokens(T, B, vocab, pad, seed):
    rng = np.random.default_rng(seed)
    x = np.full((T, B), pad, dtype=np.int64)
    for b in range(B):
        n = int(rng.integers(3, T + 1))
        x[:n, b] = rng.integers(3, vocab, size=n)
        x[0, b] = 2  # start symbol (text_vae.py:105)
    return x


def main():
    sys.path.insert(0, REF)
    import text_vae  # noqa

    out = {}
    for tag, (latent, vocab, pad, T, B) in {"a": (8, 50, 0, 7, 3), "b": (32, 97, 0, 12, 5)}.items():
        torch.manual_seed(0)
        model = text_vae.DAE(latent, vocab, pad)
        fill_parameters(model)
        model.eval()
        x = torch.from_numpy(make_tokens(T, B, vocab, pad, seed=11 + T))
        inputs, targets = x[:-1], x[1:]
        with torch.no_grad():
            h = model(inputs)
            z = torch.from_numpy(np.linspace(-1, 1, B * latent, dtype=np.float32).reshape(B, latent))
            logits, (hn, cn) = model.decode(z, inputs)
            rec = model.loss_rec(logits, targets)
            acc = model.accuracy(logits, targets)
        out[f"{tag}__cfg"] = np.array([latent, vocab, pad, T, B])
        out[f"{tag}__tokens"] = x.numpy()
        out[f"{tag}__h"] = h.numpy()
        out[f"{tag}__z"] = z.numpy()
        out[f"{tag}__logits"] = logits.numpy()
        out[f"{tag}__hn"] = hn.numpy()
        out[f"{tag}__cn"] = cn.numpy()
        out[f"{tag}__loss_rec"] = rec.numpy()
        out[f"{tag}__accuracy"] = np.float32(acc)
    mu = torch.from_numpy(np.linspace(-2, 2, 24, dtype=np.float32).reshape(4, 6))
    lv = torch.from_numpy(np.linspace(-3, 1, 24, dtype=np.float32).reshape(4, 6))
    out["kl__mu"], out["kl__logvar"] = mu.numpy(), lv.numpy()
    out["kl__value"] = text_vae.loss_kl(mu, lv).numpy()
    np.savez_compressed(os.path.join(HERE, "textvae_golden.npz"), **out)
    print("wrote textvae_golden.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
