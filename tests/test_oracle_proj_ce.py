"""The numpy restatement of projection + cross-entropy (oracle/proj_ce_oracle.py)
against what the reference's own DAE produced (textvae_golden.npz: logits,
loss_rec, accuracy from src/text_vae.py run by tests/golden/make_text_golden.py),
and its gradients against torch autograd of the same expression."""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import proj_ce_oracle as O

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "textvae_golden.npz"))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_oracle_loss_and_accuracy_match_the_reference_outputs(tag):
    latent, vocab, pad, T, B = (int(v) for v in G[f"{tag}__cfg"])
    x = G[f"{tag}__tokens"]
    targets = x[1:]
    logits = G[f"{tag}__logits"]
    assert np.allclose(O.loss_rec(logits, targets, pad), G[f"{tag}__loss_rec"], rtol=2e-6)
    assert abs(O.accuracy(logits, targets, pad) - float(G[f"{tag}__accuracy"])) < 1e-6


def test_oracle_backward_matches_autograd():
    rng = np.random.default_rng(5)
    R, K, V, pad = 23, 12, 17, 0
    H = rng.standard_normal((R, K)).astype(np.float32)
    W = (0.3 * rng.standard_normal((V, K))).astype(np.float32)
    b = (0.1 * rng.standard_normal(V)).astype(np.float32)
    t = rng.integers(0, V, size=R)
    t[::4] = pad
    g = rng.standard_normal(R).astype(np.float32)
    Ht, Wt, bt = (torch.tensor(a, dtype=torch.float64, requires_grad=True) for a in (H, W, b))
    loss = F.cross_entropy(Ht @ Wt.T + bt, torch.from_numpy(t), ignore_index=pad, reduction="none")
    assert np.allclose(loss.detach().numpy(), O.token_loss(O.project(H, W, b), t, pad), rtol=1e-12, atol=1e-12)
    (loss * torch.from_numpy(g).double()).sum().backward()
    dH, dW, db = O.backward(H, W, b, t, pad, g)
    assert np.allclose(dH, Ht.grad.numpy(), atol=1e-12)
    assert np.allclose(dW, Wt.grad.numpy(), atol=1e-12)
    assert np.allclose(db, bt.grad.numpy(), atol=1e-12)
