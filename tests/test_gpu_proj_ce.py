"""Vocabulary projection + token cross-entropy through the C ABI
(gtg_proj_ce_fwd / gtg_proj_ce_bwd) against the numpy restatement of
text_vae.py:65,77-84 (oracle/proj_ce_oracle.py), and the text decoder's fused
step against the outputs of the reference's own DAE (textvae_golden.npz)."""
import importlib.util
import os
import sys

import numpy as np
import pytest
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import proj_ce_oracle as O

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "textvae_golden.npz"))

# fp32 tolerances: per-token loss abs 2e-5 (values ~ log V); gradients relative to the tensor's largest entry
LOSS_ATOL = 2e-5
GRAD_RTOL = 2e-5


def _case(R, K, V, pad_every, seed):
    rng = np.random.default_rng(seed)
    H = rng.standard_normal((R, K)).astype(np.float32)
    W = (rng.standard_normal((V, K)) / np.sqrt(K)).astype(np.float32)
    b = (0.1 * rng.standard_normal(V)).astype(np.float32)
    t = rng.integers(1, V, size=R)
    if pad_every:
        t[rng.random(R) < 1.0 / pad_every] = 0
    g = rng.standard_normal(R).astype(np.float32)
    return H, W, b, t, g


def _run(H, W, b, t, g, mode, pad=0):
    from gtg_vae_amd import functional as Fn

    Ht = torch.tensor(H, device="cuda", requires_grad=True)
    Wt = torch.tensor(W, device="cuda", requires_grad=True)
    bt = torch.tensor(b, device="cuda", requires_grad=True) if b is not None else None
    loss, pred = Fn.proj_ce(Ht, Wt, bt, torch.tensor(t, device="cuda"), pad, mode)
    (loss * torch.tensor(g, device="cuda")).sum().backward()
    return (loss.detach().cpu().numpy(), pred.cpu().numpy(), Ht.grad.cpu().numpy(), Wt.grad.cpu().numpy(),
            bt.grad.cpu().numpy() if bt is not None else None)


def _close(got, want, rtol):
    scale = np.abs(want).max() + 1e-30
    return np.abs(got - want).max() <= rtol * scale


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "f16x2"])
@pytest.mark.parametrize("R,K,V,pad_every", [
    (37, 24, 50, 3),        # ragged everything, scalar row tails
    (300, 64, 1000, 4),
    (1, 8, 5, 0),           # one row, no pad
    (520, 128, 4099, 2),    # vocab not a multiple of 4, half the rows pad
    (2048, 256, 30522, 3),  # the reference's vocabulary, rows held in registers
])
def test_proj_ce_matches_oracle(R, K, V, pad_every, mode):
    H, W, b, t, g = _case(R, K, V, pad_every, seed=R + V)
    loss, pred, dH, dW, db = _run(H, W, b, t, g, mode)
    logits = O.project(H, W, b)
    want = O.token_loss(logits, t, 0)
    assert np.abs(loss - want).max() <= LOSS_ATOL
    keep = t != 0
    # arg-max: equal, or a numerical tie in the oracle's own logits
    top = O.argmax_tokens(logits)
    bad = keep & (pred != top)
    if bad.any():
        rows = np.nonzero(bad)[0]
        assert np.all(np.abs(logits[rows, pred[rows]] - logits[rows, top[rows]]) < 1e-5)
    assert np.all(pred[~keep] == -1)
    wH, wW, wb = O.backward(H, W, b, t, 0, g)
    assert _close(dH, wH, GRAD_RTOL) and _close(dW, wW, GRAD_RTOL) and _close(db, wb, GRAD_RTOL)
    assert np.all(dH[~keep] == 0)


def test_proj_ce_vocab_beyond_the_register_cache_and_no_bias():
    H, W, b, t, g = _case(64, 32, 40000, 5, seed=9)
    loss, pred, dH, dW, _ = _run(H, W, None, t, g, "f32")
    logits = O.project(H, W, None)
    assert np.abs(loss - O.token_loss(logits, t, 0)).max() <= LOSS_ATOL
    wH, wW, _ = O.backward(H, W, None, t, 0, g)
    assert _close(dH, wH, GRAD_RTOL) and _close(dW, wW, GRAD_RTOL)


def test_proj_ce_all_rows_pad_gives_zero_loss_and_gradients():
    H, W, b, t, g = _case(70, 16, 33, 0, seed=2)
    t[:] = 0
    loss, pred, dH, dW, db = _run(H, W, b, t, g, "f32")
    assert np.all(loss == 0) and np.all(pred == -1)
    assert np.all(dH == 0) and np.all(dW == 0) and np.all(db == 0)


def test_proj_ce_bf16_mode_meets_the_bf16_tolerances():
    H, W, b, t, g = _case(1024, 256, 8192, 4, seed=4)
    loss, pred, dH, dW, db = _run(H, W, b, t, g, "bf16")
    want = O.token_loss(O.project(H, W, b), t, 0)
    assert abs(loss.sum() - want.sum()) <= 1e-3 * abs(want.sum())
    wH, wW, wb = O.backward(H, W, b, t, 0, g)
    for got, ref in ((dH, wH), (dW, wW), (db, wb)):
        cos = float((got * ref).sum() / (np.linalg.norm(got) * np.linalg.norm(ref)))
        assert cos >= 0.999


def _fill_parameters():
    spec = importlib.util.spec_from_file_location("make_text_golden", os.path.join(HERE, "golden", "make_text_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.fill_parameters


@pytest.mark.parametrize("tag", ["a", "b"])
def test_decode_loss_matches_reference_dae_outputs(tag):
    """DAE.decode_loss (stock LSTM + the HIP projection/cross-entropy) against loss_rec and accuracy
    computed by the reference's DAE.decode / loss_rec / accuracy on the same weights and tokens."""
    from gtg_vae_amd.text_vae import DAE

    latent, vocab, pad, T, B = (int(v) for v in G[f"{tag}__cfg"])
    model = DAE(latent, vocab, pad)
    _fill_parameters()(model)
    model.eval().cuda()
    x = torch.from_numpy(G[f"{tag}__tokens"]).cuda()
    inputs, targets = x[:-1], x[1:]
    with torch.no_grad():
        tok_loss, pred, _ = model.decode_loss(torch.from_numpy(G[f"{tag}__z"]).cuda(), inputs, targets, matrix_mode="f32")
    assert np.allclose(tok_loss.sum(dim=0).cpu().numpy(), G[f"{tag}__loss_rec"], rtol=1e-5)
    assert abs(model.accuracy_from_pred(pred, targets) - float(G[f"{tag}__accuracy"])) < 1e-6
