"""The graph-VAE CPU oracle against (a) the golden vectors produced by the
reference's own graph_vae.py and (b) its independent dense GCN form."""
import os

import numpy as np
import pytest
import torch

from oracle import graphvae_oracle as O


@pytest.fixture(scope="module")
def gold(golden_dir):
    return np.load(os.path.join(golden_dir, "graphvae_golden.npz"))


def _dec_params(gold, tag):
    return {f"decoder.{k.split('__dec__')[1]}": torch.from_numpy(gold[k]) for k in gold.files if k.startswith(f"{tag}__dec__")}


@pytest.mark.parametrize("tag", ["a", "b"])
def test_decoder_and_loss_match_reference(gold, tag):
    p = {k: v.clone().requires_grad_(True) for k, v in _dec_params(gold, tag).items()}
    z = torch.from_numpy(gold[f"{tag}__z"]).requires_grad_(True)
    out = O.decode(p, z)
    assert torch.allclose(out, torch.from_numpy(gold[f"{tag}__out"]), rtol=1e-6, atol=1e-7)
    adj = torch.from_numpy(gold[f"{tag}__adj"])
    assert np.array_equal(O.triu_targets(adj).numpy(), gold[f"{tag}__vec"])
    loss = O.recon_loss(out, adj)
    assert loss.item() == pytest.approx(float(gold[f"{tag}__bce_mean"]), rel=1e-6)
    loss.backward()
    assert torch.allclose(z.grad, torch.from_numpy(gold[f"{tag}__dz"]), rtol=1e-5, atol=1e-8)
    for k, v in p.items():
        want = torch.from_numpy(gold[f"{tag}__grad__{k.replace('decoder.', '')}"])
        assert torch.allclose(v.grad, want, rtol=1e-5, atol=1e-8), k


def test_bce_clamp_matches_reference(gold):
    p, t = torch.from_numpy(gold["edge__p"]), torch.from_numpy(gold["edge__t"])
    each = torch.nn.functional.binary_cross_entropy(p, t, reduction="none")
    assert np.array_equal(each.numpy(), gold["edge__bce_each"])
    assert each[0].item() == 100.0 and each[1].item() == 100.0


@pytest.mark.parametrize("B,N,d", [(5, 10, 8), (3, 32, 16)])
def test_scatter_gcn_equals_dense_gcn(B, N, d):
    from gtg_vae_amd.data import synth_graph_batch

    batch = synth_graph_batch(B, N, seed=3)
    params = O.init_params(N, d, 32, N, seed=1)
    params["encoder.conv1.bias"] = torch.randn(2 * d) * 0.1
    params["encoder.conv2.bias"] = torch.randn(d) * 0.1
    a = O.encode(params, None, batch.edge_index, B, N)
    b = O.encode_dense(params, None, batch.edge_index, B, N)
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-6)
    # general (non-identity) node features as well
    x = torch.randn(B * N, N)
    a = O.encode(params, x, batch.edge_index, B, N)
    b = O.encode_dense(params, x, batch.edge_index, B, N)
    assert torch.allclose(a, b, rtol=1e-5, atol=1e-5)


def test_adam_matches_torch_optim():
    torch.manual_seed(0)
    p0 = torch.randn(257)
    ref = p0.clone().requires_grad_(True)
    opt = torch.optim.Adam([ref], lr=1e-3, betas=(0.5, 0.999))
    p, m, v = p0.clone(), torch.zeros(257), torch.zeros(257)
    for step in range(1, 6):
        g = torch.randn(257)
        ref.grad = g.clone()
        opt.step()
        O.adam_step(p, g, m, v, step)
        assert torch.allclose(p, ref.detach(), rtol=1e-6, atol=1e-7)
