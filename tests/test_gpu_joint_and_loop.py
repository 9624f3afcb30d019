"""GPU tests of the layers around the graph hot path: the joint text+graph
forward (gtvae.py:95-126), the named benchmark configurations A and B as
whole training steps, and the train()/dev() loop."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


def _relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


def _tokens(B, T, vocab, seed):
    from make_text_golden import make_tokens

    return torch.from_numpy(make_tokens(T, B, vocab, 0, seed)).t().contiguous()  # (B, T) as train.py feeds dec_seq


def test_joint_forward_matches_composition_of_reference_pieces():
    """text=True: joint_h = (text_h + graph_h)/2, loss = kl_w*KL + text_recon, graph_recon only reported
    (gtvae.py:107-124).  The graph pieces come from the CPU oracle, the text pieces from the same DAE
    weights run on the CPU (the DAE is pinned to the reference's outputs in test_text_branch.py)."""
    from gtg_vae_amd import GTVAE
    from gtg_vae_amd.data import synth_graph_batch
    from gtg_vae_amd.text_vae import DAE
    from make_text_golden import fill_parameters
    from oracle import graphvae_oracle as O

    B, N, d, hid, vocab, T = 6, 10, 32, 64, 60, 9
    torch.manual_seed(0)
    model = GTVAE(N, d, hid, N, vocab, 0, text=True)
    fill_parameters(model.text_vae, scale=0.05)
    model.eval()  # no dropout / word noise: deterministic
    sd = {k: v.clone() for k, v in model.state_dict().items()}
    batch = synth_graph_batch(B, N, seed=5)
    dec_seq = _tokens(B, T, vocab, 3)
    eps = torch.randn(B, d, generator=torch.Generator().manual_seed(2))
    kl_w = 0.4

    # ---- composition on the CPU
    gp = {k.replace("graph_vae.", ""): v for k, v in sd.items() if not k.startswith("text_vae.")}
    graph_h = O.encode(gp, None, batch.edge_index, B, N)
    dae = DAE(d, vocab, 0)
    dae.load_state_dict({k[len("text_vae."):]: v for k, v in sd.items() if k.startswith("text_vae.")})
    dae.eval()
    inputs = dec_seq.t()
    targets = torch.cat((dec_seq[:, 1:], torch.zeros(B, 1, dtype=dec_seq.dtype)), 1).long().t()
    with torch.no_grad():
        text_h = dae.autoenc(inputs, targets)
        joint_h = (text_h + graph_h) / 2
        z, mu, logvar, kl = O.latent(gp, joint_h, eps)
        logits, _ = dae.decode(z, inputs)
        text_rec = dae.loss_rec(logits, targets).mean()
        text_acc = dae.accuracy(logits, targets)
        out = O.decode(gp, z)
        graph_rec = O.recon_loss(out, batch.adj_padded)
    want_loss = kl_w * kl + text_rec

    # ---- the module on the GPU
    model = model.cuda()
    gb = batch.to("cuda")
    with torch.no_grad():
        res = model(gb.dense_x(), gb.edge_index, gb.batch, gb.adj_padded, gb.gold_edges, dec_seq.cuda(), kl_w,
                    eps=eps.cuda())
    assert len(res) == 10 and res[7:] == (0, 0, 0)
    loss, acc, gkl, grec, trec = res[0], res[1], res[2], res[3], res[4]
    assert abs(gkl.item() - kl.item()) <= 2e-5 * abs(kl.item())
    assert abs(trec.item() - text_rec.item()) <= 2e-5 * abs(text_rec.item())
    assert abs(grec.item() - graph_rec.item()) <= 2e-5 * abs(graph_rec.item())
    assert abs(loss.item() - want_loss.item()) <= 2e-5 * abs(want_loss.item())
    assert abs(acc - text_acc) < 1e-6


@pytest.mark.parametrize("mode", ["f32", "bf16x3", "f16x2"])
@pytest.mark.parametrize("name,B,N,d,hid", [("A", 64, 10, 32, 512), ("B", 256, 32, 256, 512), ("C", 512, 64, 512, 512)])
def test_named_config_training_step_matches_oracle(name, B, N, d, hid, mode):
    """Reference defaults (A), the mid-size configuration (B) and the headline configuration at full size
    (C: 512 graphs, N=64, d=512): losses and every gradient, eager and graph replay, against the oracle on
    the same weights, batch and eps."""
    from gtg_vae_amd.data import synth_graph_batch
    from gtg_vae_amd.trainer import GraphVAETrainer
    from oracle import graphvae_oracle as O

    params = O.init_params(N, d, hid, N, seed=0)
    tr = GraphVAETrainer(N, d, hid, N, device="cuda", init=params, matrix_mode=mode)
    batch = synth_graph_batch(B, N, seed=1234)
    eps = torch.randn(B, d, generator=torch.Generator().manual_seed(9))

    def oracle(pool_index=None):
        p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
        loss, rec, kl, _ = O.graph_step_loss(p, None, batch.edge_index, batch.adj_padded, eps, 0.5, B, N, pool_index)
        loss.backward()
        return p, rec, kl

    p, rec, kl = oracle()
    for use_graph in (False, True):
        tr.load_state_dict(params)
        res = tr.step(batch.to("cuda"), eps.cuda(), kl_w=0.5, update=False, use_graph=use_graph)
        assert tr.edges_ok()
        assert abs(res["recon"].item() - rec.item()) <= 1e-5 * abs(rec.item())
        assert abs(res["kl"].item() - kl.item()) <= 1e-5 * abs(kl.item()) + 1e-7
        # The max-pool's winners against the oracle's.  Where two rows of a graph tie to within rounding the two
        # implementations may pick different ones, and each such flip moves one graph's share of one column's
        # gradient to another row.  A flip is legitimate only on such a tie, and only a handful may occur; the
        # gradients are then compared with the oracle routed through the same winners, at the same 1e-4.
        with torch.no_grad():
            _, nodes = O.encode(params, None, batch.edge_index, B, N, return_nodes=True)
        won = tr.argmax[:B].long().cpu()
        top2 = nodes.topk(min(2, N), dim=1).values
        flipped = (nodes.gather(1, won.view(B, 1, d)).squeeze(1) != top2[:, 0]).nonzero()
        assert len(flipped) <= 8, (name, mode, len(flipped))
        for b, c in flipped.tolist():
            lead = (top2[b, 0, c] - nodes[b, won[b, c], c]).abs().item()
            assert lead <= 1e-6 * max(abs(top2[b, 0, c].item()), 1e-30), (name, mode, b, c, lead)
        want = oracle(won)[0] if len(flipped) else p
        for k in want:
            assert _relerr(tr.named_grads()[k], want[k].grad) < 1e-4, (name, mode, use_graph, k)


@pytest.mark.parametrize("case", range(14))
def test_random_awkward_shapes_match_oracle(case):
    """Odd sizes end to end (N from 1 to 64, single-graph batches, d or hid of 1, non-multiples of every
    tile size): losses and all gradients of the fused step against the oracle, eager and graph replay."""
    from gtg_vae_amd.data import synth_graph_batch
    from gtg_vae_amd.trainer import GraphVAETrainer
    from oracle import graphvae_oracle as O

    rng = np.random.default_rng(1000 + case)
    N = int(rng.choice([1, 2, 3, 5, 10, 17, 31, 33, 48, 63, 64]))
    B = int(rng.choice([1, 2, 3, 7, 33, 130, 257]))
    d = int(rng.choice([1, 2, 3, 12, 30, 64, 100]))
    hid = int(rng.choice([1, 5, 64, 96, 130]))
    params = O.init_params(N, d, hid, N, seed=case)
    for k in params:
        if k.endswith("bias"):
            params[k] = 0.1 * torch.randn(params[k].shape, generator=torch.Generator().manual_seed(case))
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    batch = synth_graph_batch(B, N, seed=100 + case)
    eps = torch.randn(B, d, generator=torch.Generator().manual_seed(case))
    kl_w = float(rng.uniform(0, 1))
    loss, rec, kl, _ = O.graph_step_loss(p, None, batch.edge_index, batch.adj_padded, eps, kl_w, B, N)
    loss.backward()
    tr = GraphVAETrainer(N, d, hid, N, device="cuda", init=params)
    for use_graph in (False, True):
        res = tr.step(batch.to("cuda"), eps.cuda(), kl_w=kl_w, update=False, use_graph=use_graph)
        assert tr.edges_ok()
        assert abs(res["recon"].item() - rec.item()) <= 1e-5 * abs(rec.item())
        assert abs(res["kl"].item() - kl.item()) <= 1e-5 * abs(kl.item()) + 1e-7
        for k in p:
            if p[k].grad.abs().max() > 0:
                assert _relerr(tr.named_grads()[k], p[k].grad) < 1e-4, (B, N, d, hid, use_graph, k)


def test_bf16_matrix_mode_meets_the_bf16_tolerances():
    """Plain bf16 operands in the large encoder GEMMs (config B, where they run on the 128x128 tiles): loss
    within 1e-3 relative, every gradient at cosine similarity >= 0.999 with the fp32 oracle."""
    from gtg_vae_amd.data import synth_graph_batch
    from gtg_vae_amd.trainer import GraphVAETrainer
    from oracle import graphvae_oracle as O

    B, N, d, hid = 256, 32, 256, 512
    params = O.init_params(N, d, hid, N, seed=0)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    batch = synth_graph_batch(B, N, seed=1234)
    eps = torch.randn(B, d, generator=torch.Generator().manual_seed(9))
    loss, rec, kl, _ = O.graph_step_loss(p, None, batch.edge_index, batch.adj_padded, eps, 0.5, B, N)
    loss.backward()
    tr = GraphVAETrainer(N, d, hid, N, device="cuda", init=params, matrix_mode="bf16")
    res = tr.step(batch.to("cuda"), eps.cuda(), kl_w=0.5, update=False)
    assert abs(res["recon"].item() - rec.item()) <= 1e-3 * abs(rec.item())
    assert abs(res["kl"].item() - kl.item()) <= 1e-3 * abs(kl.item()) + 1e-6
    exact = True
    for k in p:
        g, w = tr.named_grads()[k].double().cpu().flatten(), p[k].grad.double().flatten()
        cos = torch.dot(g, w) / (g.norm() * w.norm() + 1e-30)
        assert cos.item() >= 0.999, (k, cos.item())
        exact = exact and _relerr(tr.named_grads()[k], p[k].grad) < 1e-6
    assert not exact        # the coarse pipe really ran
