"""The text-branch oracle (explicit LSTM recurrences, oracle/textvae_oracle.py) against the vectors the
reference's own src/text_vae.py produced (tests/golden/textvae_golden.npz, make_text_golden.py), and
against torch.nn.LSTM on random inputs."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
G = np.load(os.path.join(HERE, "golden", "textvae_golden.npz"))


def _params(latent, vocab, pad):
    """The hash-filled DAE weights the goldens were generated with, as a plain dict."""
    from gtg_vae_amd.text_vae import DAE   # layer shapes / state_dict names only
    from make_text_golden import fill_parameters

    m = DAE(latent, vocab, pad)
    fill_parameters(m)
    return {k: v.clone() for k, v in m.state_dict().items()}


@pytest.mark.parametrize("tag", ["a", "b"])
def test_text_oracle_matches_reference_outputs(tag):
    from oracle import textvae_oracle as T

    latent, vocab, pad, Tn, B = (int(v) for v in G[f"{tag}__cfg"])
    p = _params(latent, vocab, pad)
    x = torch.from_numpy(G[f"{tag}__tokens"])
    inputs, targets = x[:-1], x[1:]
    with torch.no_grad():
        h = T.encode(p, inputs)
        logits, (hn, cn) = T.decode(p, torch.from_numpy(G[f"{tag}__z"]), inputs)
        rec = T.loss_rec(logits, targets, pad)
        acc = T.accuracy(logits, targets, pad)
    assert np.allclose(h.numpy(), G[f"{tag}__h"], rtol=1e-5, atol=1e-6)
    assert np.allclose(logits.numpy(), G[f"{tag}__logits"], rtol=1e-5, atol=1e-5)
    assert np.allclose(hn.numpy(), G[f"{tag}__hn"][0], rtol=1e-5, atol=1e-6)
    assert np.allclose(cn.numpy(), G[f"{tag}__cn"][0], rtol=1e-5, atol=1e-6)
    assert np.allclose(rec.numpy(), G[f"{tag}__loss_rec"], rtol=1e-5)
    assert abs(acc - float(G[f"{tag}__accuracy"])) < 1e-6


@pytest.mark.parametrize("reverse", [False, True])
def test_explicit_recurrence_equals_nn_lstm_with_gradients(reverse):
    from oracle import textvae_oracle as T

    torch.manual_seed(1)
    Tn, B, I, H = 6, 3, 5, 7
    ref = torch.nn.LSTM(I, H, 1, bidirectional=True)
    x = torch.randn(Tn, B, I, requires_grad=True)
    out, (hn, cn) = ref(x)
    sfx = "_reverse" if reverse else ""
    w = [getattr(ref, n + "_l0" + sfx) for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")]
    x2 = x.detach().clone().requires_grad_(True)
    w2 = [t.detach().clone().requires_grad_(True) for t in w]
    o2, h2, c2 = T.lstm(x2, *w2, reverse=reverse)
    half = out[:, :, H:] if reverse else out[:, :, :H]
    assert torch.allclose(o2, half, atol=1e-6)
    assert torch.allclose(h2, hn[1 if reverse else 0], atol=1e-6) and torch.allclose(c2, cn[1 if reverse else 0], atol=1e-6)
    half.square().sum().backward()
    o2.square().sum().backward()
    for a, b in zip(w, w2):
        assert torch.allclose(a.grad, b.grad, atol=1e-5)


@pytest.mark.parametrize("tag", ["small", "batch", "long", "keep_all", "drop_most"])
def test_word_drop_oracle_reproduces_the_reference(tag):
    """noise_golden.npz holds tokens, the keep flags the reference drew (noise.py:16) and its output."""
    import os

    import numpy as np

    from oracle import textvae_oracle as TO

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "noise_golden.npz"))
    pad = int(g[tag + "__meta"][0])
    got = TO.word_drop(g[tag + "__x"], g[tag + "__keep"], pad)
    assert np.array_equal(got.numpy(), g[tag + "__out"])
