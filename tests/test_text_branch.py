"""Text branch (stock-PyTorch mirror of the reference's DAE) against vectors
produced by running the reference's own src/text_vae.py
(tests/golden/make_text_golden.py).  CPU: the branch is plain torch modules."""
import importlib.util
import os

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
G = np.load(os.path.join(HERE, "golden", "textvae_golden.npz"))


def _fill_parameters():
    spec = importlib.util.spec_from_file_location("make_text_golden", os.path.join(HERE, "golden", "make_text_golden.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.fill_parameters


@pytest.mark.parametrize("tag", ["a", "b"])
def test_dae_eval_matches_reference_outputs(tag):
    from gtg_vae_amd.text_vae import DAE

    latent, vocab, pad, T, B = (int(v) for v in G[f"{tag}__cfg"])
    model = DAE(latent, vocab, pad)
    _fill_parameters()(model)
    model.eval()
    x = torch.from_numpy(G[f"{tag}__tokens"])
    inputs, targets = x[:-1], x[1:]
    with torch.no_grad():
        h = model(inputs)
        logits, (hn, cn) = model.decode(torch.from_numpy(G[f"{tag}__z"]), inputs)
        rec = model.loss_rec(logits, targets)
        acc = model.accuracy(logits, targets)
    assert np.allclose(h.numpy(), G[f"{tag}__h"], rtol=1e-5, atol=1e-6)
    assert np.allclose(logits.numpy(), G[f"{tag}__logits"], rtol=1e-5, atol=1e-5)
    assert np.allclose(hn.numpy(), G[f"{tag}__hn"], rtol=1e-5, atol=1e-6)
    assert np.allclose(cn.numpy(), G[f"{tag}__cn"], rtol=1e-5, atol=1e-6)
    assert np.allclose(rec.numpy(), G[f"{tag}__loss_rec"], rtol=1e-5)
    assert abs(acc - float(G[f"{tag}__accuracy"])) < 1e-6


def test_dae_state_dict_keys_are_the_reference_ones():
    from gtg_vae_amd.text_vae import DAE

    keys = set(DAE(8, 50, 0).state_dict().keys())
    want = {"embed.weight", "proj.weight", "proj.bias", "hgen.weight", "hgen.bias", "z2emb.weight", "z2emb.bias"}
    want |= {f"E.{w}_l0{s}" for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh") for s in ("", "_reverse")}
    want |= {f"G.{w}_l0" for w in ("weight_ih", "weight_hh", "bias_ih", "bias_hh")}
    assert keys == want


@pytest.mark.gpu
def test_loss_kl_matches_reference_value():
    from gtg_vae_amd.gtvae import GTVAE

    mu, lv = torch.from_numpy(G["kl__mu"]).cuda(), torch.from_numpy(G["kl__logvar"]).cuda()
    got = GTVAE.loss_kl(None, mu, lv)
    assert abs(float(got) - float(G["kl__value"])) <= 1e-6 * abs(float(G["kl__value"]))


def test_training_mode_applies_word_drop_only_to_non_start_tokens():
    from gtg_vae_amd.text_vae import DAE

    torch.manual_seed(3)
    model = DAE(8, 97, 0)
    model.train()
    x = torch.from_numpy(G["b__tokens"])[:-1]
    h1 = model(x)
    assert h1.shape == (x.shape[1], 8) and torch.isfinite(h1).all()
