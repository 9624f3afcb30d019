"""The joint graph+text step (gtvae.py:95-126, train.py:290-314) on the GPU against the CPU oracles:
forward AND backward of GTVAE(text=True), forward_1, and JointTrainer's flat-buffer Adam steps.  The text
oracle writes the LSTM recurrences out explicitly and is pinned to the reference's own text_vae.py outputs
(tests/test_oracle_textvae.py); the graph oracle is oracle/graphvae_oracle.py.  Dropout and word noise are
switched off throughout (the only random input left, eps, is supplied)."""
import os
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def _relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


def _setup(B, N, d, hid, vocab, T, seed, train_graph_recon):
    """(state dict on the CPU, collated batch with sentences, eps)."""
    from gtg_vae_amd import GTVAE
    from gtg_vae_amd.data import collate, synth_graph_list
    from make_text_golden import fill_parameters

    torch.manual_seed(seed)
    model = GTVAE(N, d, hid, N, vocab, 0, text=True, train_graph_recon=train_graph_recon)
    fill_parameters(model.text_vae, scale=0.05)
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    batch = collate(synth_graph_list(B, N, seed=seed + 1, vocab=vocab, max_len=T), N)
    eps = torch.randn(B, d, generator=torch.Generator().manual_seed(seed + 2))
    return sd, batch, eps


CONFIGS = {
    # name: B, N, d, hid, vocab, T
    "small": (6, 10, 32, 64, 60, 9),
    "mid": (33, 17, 48, 96, 1000, 21),
    # BASELINE configs[4] / configs[1] sizes: batch 256, N = 32, d = 256, sentences up to 40 tokens, bert vocabulary
    "full": (256, 32, 256, 512, 30522, 40),
}


@pytest.mark.parametrize("train_graph_recon", [False, True])
@pytest.mark.parametrize("name", ["small", "mid", "full"])
def test_joint_forward_and_backward_match_oracle(name, train_graph_recon):
    from gtg_vae_amd.joint_trainer import JointTrainer
    from oracle import textvae_oracle as TO

    B, N, d, hid, vocab, T = CONFIGS[name]
    if name == "full" and not train_graph_recon:
        pytest.skip("the full size runs once, with every parameter receiving a gradient")
    sd, batch, eps = _setup(B, N, d, hid, vocab, T, 3, train_graph_recon)
    kl_w = 0.4
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    want = TO.joint_loss(p, batch.edge_index, batch.adj_padded, batch.dec_seq, eps, kl_w, B, N, 0, train_graph_recon)
    want["loss"].backward()

    tr = JointTrainer(N, d, hid, N, vocab, 0, device="cuda", init=sd, train_graph_recon=train_graph_recon)
    tr.deterministic()
    gb = batch.to("cuda")
    res = tr.step(gb, gb.dec_seq, eps.cuda(), kl_w, update=False)
    for k in ("loss", "kl", "text_recon", "graph_recon"):
        assert abs(res[k].item() - want[k].item()) <= 1e-5 * abs(want[k].item()) + 1e-7, (k, res[k].item(), want[k].item())
    assert abs(res["text_acc"].item() - want["text_acc"]) < 1e-6
    grads = tr.named_grads()
    assert set(grads) == set(p)
    for k, v in p.items():
        if v.grad is None:      # graph decoder when its loss is only reported (gtvae.py:124)
            assert k.startswith("graph_vae.decoder.") and not train_graph_recon
            assert float(grads[k].abs().max()) == 0.0
            continue
        assert _relerr(grads[k], v.grad) < 1e-4, (name, k, _relerr(grads[k], v.grad))


@pytest.mark.parametrize("name", ["small", "mid"])
def test_forward_1_matches_oracle(name):
    """GTVAE.forward_1 (gtvae.py:77-91): the text-only VAE step through the shared latent head."""
    from gtg_vae_amd import GTVAE
    from oracle import textvae_oracle as TO

    B, N, d, hid, vocab, T = CONFIGS[name]
    sd, batch, eps = _setup(B, N, d, hid, vocab, T, 5, False)
    inputs, targets = TO.shift_targets(batch.dec_seq)
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    want_loss, want_acc = TO.forward_1(p, inputs, targets, eps)
    want_loss.backward()
    model = GTVAE(N, d, hid, N, vocab, 0, text=True)
    model.load_state_dict(sd)
    model = model.cuda().train()    # training mode with dropout and word drop switched off: deterministic, and
    model.text_vae.drop.p = 0.0     # the stock LSTM has no backward in evaluation mode
    model.text_vae.word_drop = 0.0
    loss, acc = model.forward_1(inputs.cuda(), targets.cuda(), eps=eps.cuda())
    loss.backward()
    assert abs(loss.item() - want_loss.item()) <= 1e-5 * abs(want_loss.item())
    assert abs(acc - want_acc) < 1e-6
    for k, v in model.named_parameters():
        if p[k].grad is None:
            assert v.grad is None or float(v.grad.abs().max()) == 0.0, k
        else:
            assert _relerr(v.grad, p[k].grad) < 1e-4, k


def test_joint_trainer_adam_steps_follow_oracle():
    """Three optimiser steps through the flat buffers (one gtg_adam_step over every parameter, text ones
    included).  Per step: the gradients the step consumed against the oracle's (1e-4), then the parameters
    against Adam(1e-3, (0.5, 0.999)) applied on the CPU to those same gradients (1e-6: Adam divides by
    sqrt(v), so entries whose gradient is rounding noise would otherwise dominate a parameter comparison).
    The module's parameters are views of the flat buffer, so its state_dict follows."""
    from gtg_vae_amd.data import collate, synth_graph_list
    from gtg_vae_amd.joint_trainer import JointTrainer
    from oracle import graphvae_oracle as O
    from oracle import textvae_oracle as TO

    B, N, d, hid, vocab, T = CONFIGS["small"]
    sd, _, _ = _setup(B, N, d, hid, vocab, T, 7, True)
    tr = JointTrainer(N, d, hid, N, vocab, 0, device="cuda", init=sd, train_graph_recon=True)
    tr.deterministic()
    p = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    for step in range(1, 4):
        batch = collate(synth_graph_list(B, N, seed=20 + step, vocab=vocab, max_len=T), N)
        eps = torch.randn(B, d, generator=torch.Generator().manual_seed(step))
        gb = batch.to("cuda")
        tr.step(gb, gb.dec_seq, eps.cuda(), 0.3 * step, update=True)
        for t in p.values():
            t.grad = None
        TO.joint_loss(p, batch.edge_index, batch.adj_padded, batch.dec_seq, eps, 0.3 * step, B, N, 0, True)["loss"].backward()
        used = {k: g.detach().cpu().clone() for k, g in tr.named_grads().items()}   # what this step's Adam consumed
        with torch.no_grad():
            for k in p:
                assert _relerr(used[k], p[k].grad) < 1e-4, (step, k)
                O.adam_step(p[k], used[k], m[k], v2[k], step)
        for k in p:
            assert (tr.named_params()[k].cpu() - p[k].detach()).abs().max().item() < 1e-6, (step, k)
    got = tr.state_dict()
    assert set(got) == set(p)
    for k in p:
        assert torch.equal(got[k], tr.named_params()[k])
        assert got[k].data_ptr() != tr.named_params()[k].data_ptr()            # state_dict hands out copies
    mod = dict(tr.model.named_parameters())
    assert all(mod[k].data_ptr() == tr.named_params()[k].data_ptr() for k in mod)   # one copy of every weight
