"""GPU parity of the graph-VAE hot path against the CPU oracle and the golden
vectors: adjacency, encoder, latent head, decoder, BCE, full step gradients,
Adam, determinism, HIP-graph replay.

Tolerances (fp32 kernels): loss rel <= 1e-5, gradients max-abs-error /
max-abs-value <= 1e-4, as stated for the fp32 path.
"""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

CONFIGS = [(8, 10, 32, 64), (5, 32, 64, 96), (3, 64, 128, 512), (64, 10, 32, 512)]


def _relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


def _oracle_params(N, d, hid, seed=1, rand_bias=True):
    from oracle import graphvae_oracle as O

    p = O.init_params(N, d, hid, N, seed=seed)
    if rand_bias:
        g = torch.Generator().manual_seed(seed + 50)
        p["encoder.conv1.bias"] = torch.randn(2 * d, generator=g) * 0.1
        p["encoder.conv2.bias"] = torch.randn(d, generator=g) * 0.1
    return p


@pytest.mark.parametrize("B,N", [(4, 10), (7, 32), (5, 64), (3, 1)])
def test_adj_build_matches_oracle(B, N):
    from gtg_vae_amd import functional as Fn
    from gtg_vae_amd.data import synth_graph_batch
    from oracle import graphvae_oracle as O

    if N == 1:
        ei = torch.zeros((2, 3), dtype=torch.long)
        ei[:, 1] = 1
        ei[:, 2] = 2
    else:
        ei = synth_graph_batch(B, N, seed=B * N).edge_index
    want = O.gcn_dense_adj(ei, B, N)
    got = Fn.build_adj(ei.cuda(), B, N, check_edges=True)
    assert torch.allclose(got.cpu(), want, rtol=1e-6, atol=1e-7)


def test_adj_build_extents_and_reusable_counts_scratch():
    """gtg_adj_build leaves its counts scratch all zero (two different batches through the same
    buffers give two right answers) and reports each graph's extent: 1 + the highest node index a
    non-self edge touches, 0 without edges."""
    from gtg_vae_amd import _lib
    from gtg_vae_amd._lib import check, ptr, stream_ptr
    from gtg_vae_amd.data import synth_graph_batch
    from oracle import graphvae_oracle as O

    B, N = 9, 12
    counts = torch.zeros(B * N * N, dtype=torch.int32, device="cuda")
    err = torch.zeros(1, dtype=torch.int32, device="cuda")
    adj = torch.empty(B, N, N, device="cuda")
    ext = torch.empty(B, dtype=torch.int32, device="cuda")
    for seed in (3, 4):
        ei = synth_graph_batch(B, N, seed=seed).edge_index
        if seed == 4:  # graph 0 loses its edges, graph 1 keeps self loops only
            keep = (ei[0] // N > 1)
            ei = torch.cat([ei[:, keep], torch.tensor([[N + 2, N + 5], [N + 2, N + 5]])], 1)
        check(_lib.lib().gtg_adj_build(ptr(ei.cuda()), ei.shape[1], B, N, ptr(counts), ptr(err), ptr(adj), ptr(ext),
                                       stream_ptr()), "gtg_adj_build")
        assert torch.allclose(adj.cpu(), O.gcn_dense_adj(ei, B, N), rtol=1e-6, atol=1e-7)
        assert int(counts.abs().sum()) == 0 and int(err.item()) == 0
        want = torch.zeros(B, dtype=torch.int32)
        for s_, t_ in ei.t().tolist():
            if s_ != t_:
                g = s_ // N
                want[g] = max(int(want[g]), s_ % N + 1, t_ % N + 1)
        assert torch.equal(ext.cpu(), want)


def test_adj_build_flags_cross_graph_edges():
    from gtg_vae_amd import functional as Fn
    from gtg_vae_amd._lib import GtgError

    ei = torch.tensor([[0, 11], [1, 3]], dtype=torch.long).cuda()   # 11 -> 3 crosses graphs (N=10)
    with pytest.raises(GtgError):
        Fn.build_adj(ei, 2, 10, check_edges=True)


@pytest.mark.parametrize("B,N,d,hid", CONFIGS)
@pytest.mark.parametrize("identity_x", [True, False])
def test_encoder_forward_backward(B, N, d, hid, identity_x):
    from gtg_vae_amd import functional as Fn
    from gtg_vae_amd.data import synth_graph_batch
    from oracle import graphvae_oracle as O

    batch = synth_graph_batch(B, N, seed=11)
    p = {k: v.clone().requires_grad_(True) for k, v in _oracle_params(N, d, hid).items()}
    x = None if identity_x else torch.randn(B * N, N, generator=torch.Generator().manual_seed(2))
    want = O.encode(p, x, batch.edge_index, B, N)
    up = torch.randn(B, d, generator=torch.Generator().manual_seed(3))
    (want * up).sum().backward()

    names = ["encoder.conv1.lin.weight", "encoder.conv1.bias", "encoder.conv2.lin.weight", "encoder.conv2.bias"]
    gp = [p[k].detach().cuda().requires_grad_(True) for k in names]
    adj, extents = Fn.build_adj(batch.edge_index.cuda(), B, N, return_extents=True)
    got, argmax = Fn.EncoderFn.apply(adj, None if x is None else x.cuda(), *gp, extents)
    assert _relerr(got, want) < 1e-5
    (got * up.cuda()).sum().backward()
    for k, t in zip(names, gp):
        assert _relerr(t.grad, p[k].grad) < 1e-4, k


@pytest.mark.parametrize("N,d", [(10, 16), (64, 64)])
def test_compact_encoder_edge_cases(N, d):
    """Compact-row encoder (identity features) on awkward graphs: no edges at all,
    one directed edge, edges that touch only high node indices (untouched nodes
    in between), a full chain; forward and backward against the oracle and
    against the dense path fed an explicit identity matrix."""
    from gtg_vae_amd import functional as Fn
    from oracle import graphvae_oracle as O

    B = 6
    edges = []
    edges += [(1 * N + 0, 1 * N + 1)]                                       # graph 1: one directed edge
    edges += [(2 * N + N - 3, 2 * N + N - 1), (2 * N + N - 1, 2 * N + N - 3)]  # graph 2: only high indices
    edges += [(3 * N + i, 3 * N + i + 1) for i in range(N - 1)]              # graph 3: chain, one direction
    edges += [(3 * N + i + 1, 3 * N + i) for i in range(N - 1)]
    edges += [(4 * N + 2, 4 * N + 2), (4 * N + 0, 4 * N + 0)]                # graph 4: self loops only
    edges += [(5 * N + 0, 5 * N + 4)] * 3 + [(5 * N + 4, 5 * N + 0)]          # graph 5: multiplicity
    ei = torch.tensor(edges, dtype=torch.long).t().contiguous()             # graph 0: nothing
    p = {k: v.clone().requires_grad_(True) for k, v in _oracle_params(N, d, 16, seed=12).items()}
    want = O.encode(p, None, ei, B, N)
    up = torch.randn(B, d, generator=torch.Generator().manual_seed(4))
    (want * up).sum().backward()
    names = ["encoder.conv1.lin.weight", "encoder.conv1.bias", "encoder.conv2.lin.weight", "encoder.conv2.bias"]
    adj, extents = Fn.build_adj(ei.cuda(), B, N, return_extents=True)
    outs = []
    for compact in (True, False):
        gp = [p[k].detach().cuda().requires_grad_(True) for k in names]
        x = None if compact else torch.eye(N).repeat(B, 1).cuda()
        got, am = Fn.EncoderFn.apply(adj, x, *gp, extents if compact else None)
        assert _relerr(got, want) < 1e-5, compact
        (got * up.cuda()).sum().backward()
        for k, t in zip(names, gp):
            assert _relerr(t.grad, p[k].grad) < 1e-4, (compact, k)
        outs.append((got.detach().clone(), am.clone()))
    assert torch.equal(outs[0][1], outs[1][1])          # same winning rows in both paths
    assert _relerr(outs[0][0], outs[1][0]) < 1e-6


@pytest.mark.parametrize("B,N,d", [(130, 64, 512), (300, 10, 256), (256, 33, 288), (600, 10, 512), (96, 64, 768)])
def test_conv2_weight_gradient_through_the_pool_routing(B, N, d):
    """At B*d >= 65536 the compact backward takes dW2 from the pooled gradient and the winning rows
    (one aggregated row of H1 per graph and column) instead of the dense product: awkward graphs
    cycled through the batch (empty, one edge, high indices only, chain, self loops, multi-edges,
    complete, random), a batch that is not a multiple of the 16 graph groups, a width whose last
    wave of pooled columns is partly idle (288); all four gradients against the oracle.  At these
    sizes (2048 stored rows or more, hidden width a multiple of 64) conv1's weight gradient is
    taken through the nonzeros of the adjacency rows as well: n = 64, 10 and 33 columns, one column
    per lane (2d = 512, 576) and two (2d = 1024, with n = 64 and with n = 10).  d = 768: more pooled columns
    than one block of the routing kernel takes (512), the second block's last waves idle, and the complete
    graphs there have all 64 nodes (every adjacency entry nonzero, all 64 columns walked)."""
    from gtg_vae_amd import functional as Fn
    from oracle import graphvae_oracle as O

    rng = np.random.default_rng(B + N + d)
    edges = []
    for b in range(B):
        o, kind = b * N, b % 8
        if kind == 1:
            edges += [(o, o + 1)]
        elif kind == 2:
            edges += [(o + N - 3, o + N - 1), (o + N - 1, o + N - 3)]
        elif kind == 3:
            edges += [(o + i, o + i + 1) for i in range(N - 1)] + [(o + i + 1, o + i) for i in range(N - 1)]
        elif kind == 4:
            edges += [(o + 2, o + 2), (o, o)]
        elif kind == 5:
            edges += [(o, o + 4)] * 3 + [(o + 4, o)]
        elif kind == 6:
            m = N if d > 512 else min(N, 9)
            edges += [(o + i, o + j) for i in range(m) for j in range(m) if i != j]
        elif kind == 7:
            m = int(rng.integers(2, N + 1))
            for _ in range(2 * m):
                i, j = rng.integers(0, m, size=2)
                edges += [(o + int(i), o + int(j))]
    ei = torch.tensor(edges, dtype=torch.long).t().contiguous()
    p = {k: v.clone().requires_grad_(True) for k, v in _oracle_params(N, d, 16, seed=5).items()}
    want = O.encode(p, None, ei, B, N)
    up = torch.randn(B, d, generator=torch.Generator().manual_seed(6))
    (want * up).sum().backward()
    names = ["encoder.conv1.lin.weight", "encoder.conv1.bias", "encoder.conv2.lin.weight", "encoder.conv2.bias"]
    gp = [p[k].detach().cuda().requires_grad_(True) for k in names]
    adj, extents = Fn.build_adj(ei.cuda(), B, N, return_extents=True)
    got, _ = Fn.EncoderFn.apply(adj, None, *gp, extents)
    assert _relerr(got, want) < 1e-5
    (got * up.cuda()).sum().backward()
    for k, t in zip(names, gp):
        assert _relerr(t.grad, p[k].grad) < 1e-4, k


@pytest.mark.parametrize("N,d", [(7, 6), (64, 130), (33, 5)])
def test_compact_encoder_on_dense_graphs_and_odd_widths(N, d):
    """First layer as sparse row products: complete graphs (every adjacency entry nonzero), a hidden
    width 2d that is not a multiple of the 256-column slab (260) and one that is not a multiple of 4
    (10: the dense GEMM takes over), against the oracle."""
    from gtg_vae_amd import functional as Fn
    from oracle import graphvae_oracle as O

    B = 3
    edges = [(b * N + i, b * N + j) for b in range(B) for i in range(N - b) for j in range(N - b) if i != j]
    ei = torch.tensor(edges, dtype=torch.long).t().contiguous()
    p = {k: v.clone().requires_grad_(True) for k, v in _oracle_params(N, d, 16, seed=21).items()}
    want = O.encode(p, None, ei, B, N)
    up = torch.randn(B, d, generator=torch.Generator().manual_seed(8))
    (want * up).sum().backward()
    names = ["encoder.conv1.lin.weight", "encoder.conv1.bias", "encoder.conv2.lin.weight", "encoder.conv2.bias"]
    gp = [p[k].detach().cuda().requires_grad_(True) for k in names]
    adj, extents = Fn.build_adj(ei.cuda(), B, N, return_extents=True)
    got, _ = Fn.EncoderFn.apply(adj, None, *gp, extents)
    assert _relerr(got, want) < 1e-5
    (got * up.cuda()).sum().backward()
    for k, t in zip(names, gp):
        assert _relerr(t.grad, p[k].grad) < 1e-4, k


@pytest.mark.parametrize("B,d", [(8, 32), (256, 256), (5, 7)])
def test_latent_forward_backward(B, d):
    from gtg_vae_amd import functional as Fn
    from oracle import graphvae_oracle as O

    g = torch.Generator().manual_seed(B + d)
    p = {k: v.clone().requires_grad_(True) for k, v in _oracle_params(10, d, 16).items() if k.startswith("zg_")}
    h = torch.randn(B, d, generator=g).requires_grad_(True)
    eps = torch.randn(B, d, generator=g)
    z, mu, lv, kl = O.latent(p, h, eps)
    up = torch.randn(B, d, generator=g)
    ((z * up).sum() + 0.37 * kl).backward()

    names = ["zg_mu.weight", "zg_mu.bias", "zg_var.weight", "zg_var.bias"]
    gp = [p[k].detach().cuda().requires_grad_(True) for k in names]
    hg = h.detach().cuda().requires_grad_(True)
    gz, gkl, gmu, glv = Fn.LatentFn.apply(hg, *gp, eps.cuda())
    assert _relerr(gz, z) < 1e-5 and _relerr(gmu, mu) < 1e-5 and _relerr(glv, lv) < 1e-5
    assert abs(gkl.item() - kl.item()) <= 1e-5 * abs(kl.item())
    ((gz * up.cuda()).sum() + 0.37 * gkl).backward()
    assert _relerr(hg.grad, h.grad) < 1e-4
    for k, t in zip(names, gp):
        assert _relerr(t.grad, p[k].grad) < 1e-4, k
    # stand-alone reparameterisation / KL
    mu2 = mu.detach().cuda().requires_grad_(True)
    lv2 = lv.detach().cuda().requires_grad_(True)
    z2, kl2 = Fn.ReparamKLFn.apply(mu2, lv2, eps.cuda())
    assert _relerr(z2, z) < 1e-5 and abs(kl2.item() - kl.item()) <= 1e-5 * abs(kl.item())


@pytest.mark.parametrize("tag", ["a", "b"])
def test_decoder_and_bce_match_reference_golden(golden_dir, tag):
    """Against values produced by the reference's own MLP_VAE_plain / adj_recon_loss."""
    from gtg_vae_amd import functional as Fn

    g = np.load(os.path.join(golden_dir, "graphvae_golden.npz"))
    names = ["decode_1.weight", "decode_1.bias", "decode_2.weight", "decode_2.bias"]
    gp = [torch.from_numpy(g[f"{tag}__dec__{k}"]).cuda().requires_grad_(True) for k in names]
    z = torch.from_numpy(g[f"{tag}__z"]).cuda().requires_grad_(True)
    out = Fn.DecoderFn.apply(z, *gp)
    assert torch.allclose(out.cpu(), torch.from_numpy(g[f"{tag}__out"]), rtol=1e-5, atol=1e-6)
    adj = torch.from_numpy(g[f"{tag}__adj"]).cuda()
    loss = Fn.BCELossFn.apply(out, adj)
    assert loss.item() == pytest.approx(float(g[f"{tag}__bce_mean"]), rel=1e-5)
    loss.backward()
    assert _relerr(z.grad, torch.from_numpy(g[f"{tag}__dz"])) < 1e-4
    for k, t in zip(names, gp):
        assert _relerr(t.grad, torch.from_numpy(g[f"{tag}__grad__{k}"])) < 1e-4, k


def test_bce_clamp_at_the_ends(golden_dir):
    from gtg_vae_amd import functional as Fn

    # n = 1: M = 1, one probability per "graph"
    g = np.load(os.path.join(golden_dir, "graphvae_golden.npz"))
    p, t, each = g["edge__p"], g["edge__t"], g["edge__bce_each"]
    out = torch.from_numpy(p).reshape(-1, 1).cuda()
    adj = torch.from_numpy(t).reshape(-1, 1, 1).cuda()
    loss = Fn.BCELossFn.apply(out, adj)
    assert loss.item() == pytest.approx(float(each.astype(np.float64).mean()), rel=1e-6)


@pytest.mark.parametrize("B,N,d,hid,adam", [(8, 10, 32, 64, True), (5, 32, 64, 96, False), (3, 64, 128, 512, True),
                                            (512, 64, 512, 512, True), (256, 32, 256, 512, False),
                                            (1100, 10, 32, 64, True),
                                            # fused epilogues on ragged tiles: 55 and 91 decoder outputs (not
                                            # multiples of four), batch and widths that are not multiples of 32
                                            (64, 10, 32, 512, True), (300, 13, 36, 260, False), (33, 64, 100, 256, True)])
def test_fused_head_equals_the_five_calls_bit_for_bit(B, N, d, hid, adam):
    """gtg_head_fwd_bwd (latent head + decoder + loss, forward and backward, the scalars' final sums and Adam's
    step counter riding in spare blocks of a later launch) against gtg_latent_fwd, gtg_decoder_fwd,
    gtg_bce_loss_fwd, gtg_decoder_bwd, gtg_latent_bwd and gtg_adam_step's counter: every output bit-identical
    except the reconstruction loss scalar (the loss terms and their gradient are computed in the epilogue of the
    decoder's last product and summed in another order).
    (The five calls are themselves checked against the oracle and the reference's goldens above and below.)
    1100 graphs: more rows than a riding column sum takes, the stand-alone launches run instead."""
    from gtg_vae_amd import _lib
    from gtg_vae_amd._lib import check, ptr

    L = _lib.lib()
    m = N * (N + 1) // 2
    g = torch.Generator().manual_seed(B * 7 + N)
    dev = "cuda"
    rnd = lambda *shape, s=1.0: (torch.randn(*shape, generator=g) * s).to(dev)
    h, eps = rnd(B, d), rnd(B, d)
    W = rnd(2 * d, d, s=d ** -0.5)                     # [Wmu; Wlv] back to back, as the trainer keeps them
    Wmu, Wlv = W[:d], W[d:]
    bmu, blv = rnd(d, s=0.1), rnd(d, s=0.1)
    D1, c1, D2, c2 = rnd(hid, d, s=d ** -0.5), rnd(hid, s=0.1), rnd(m, hid, s=hid ** -0.5), rnd(m, s=0.1)
    target = (torch.rand(B, N, N, generator=g) < 0.1).float().to(dev)
    klw = torch.tensor([0.37, 0.8], device=dev)        # [KL weight, shard weight of the reconstruction gradient]
    st = torch.cuda.current_stream().cuda_stream

    def outputs():
        f = lambda *shape: torch.full(shape, float("nan"), device=dev)
        return dict(mu=f(B, d), lv=f(B, d), z=f(B, d), y1=f(B, hid), out=f(B, m), dl=f(B, m), sc=f(2),
                    dW=f(2 * d, d), dbmu=f(d), dblv=f(d), dD1=f(hid, d), dc1=f(hid), dD2=f(m, hid), dc2=f(m),
                    dz=f(B, d), dh=f(B, d), adam=torch.zeros(4, device=dev))

    nb = max(L.gtg_latent_workspace_bytes(B, d), L.gtg_decoder_bwd_workspace_bytes(B, d, hid, m),
             L.gtg_bce_workspace_bytes(B), L.gtg_head_workspace_bytes(B, N, d, hid))
    ws = torch.empty(nb, dtype=torch.uint8, device=dev)
    lr, b1, b2 = 1e-3, 0.5, 0.999
    a = outputs()
    check(L.gtg_latent_fwd(ptr(h), ptr(Wmu), ptr(bmu), ptr(Wlv), ptr(blv), ptr(eps), B, d, ptr(a["mu"]), ptr(a["lv"]),
                           ptr(a["z"]), a["sc"][1:].data_ptr(), ptr(ws), nb, st), "latent_fwd")
    check(L.gtg_decoder_fwd(ptr(a["z"]), ptr(D1), ptr(c1), ptr(D2), ptr(c2), B, d, hid, m, ptr(a["y1"]), ptr(a["out"]), st),
          "decoder_fwd")
    check(L.gtg_bce_loss_fwd(ptr(a["out"]), ptr(target), B, N, ptr(a["sc"]), ptr(a["dl"]), klw[1:].data_ptr(), ptr(ws), nb,
                             st), "bce")
    check(L.gtg_decoder_bwd(ptr(a["z"]), ptr(D1), ptr(D2), ptr(a["y1"]), None, ptr(a["dl"]), None, N, 1.0, None, B, d, hid,
                            m, ptr(a["dD1"]), ptr(a["dc1"]), ptr(a["dD2"]), ptr(a["dc2"]), ptr(a["dz"]), ptr(ws), nb, st),
          "decoder_bwd")
    check(L.gtg_latent_bwd(ptr(h), ptr(Wmu), ptr(Wlv), ptr(a["mu"]), ptr(a["lv"]), ptr(eps), ptr(a["dz"]), 1.0, ptr(klw), B,
                           d, ptr(a["dW"][:d]), ptr(a["dbmu"]), ptr(a["dW"][d:]), ptr(a["dblv"]), ptr(a["dh"]), ptr(ws),
                           nb, st), "latent_bwd")
    if adam:   # the counter and bias corrections as gtg_adam_step leaves them (no elements to update)
        check(L.gtg_adam_step(None, None, None, None, 0, ptr(a["adam"]), lr, b1, b2, 1e-8, 1.0, st), "adam_step")
    torch.cuda.synchronize()
    for rep in range(2):          # twice: nothing the call leaves in its workspace may matter to the next one
        b = outputs()
        if rep == 1 and adam:
            b["adam"].copy_(a["adam"])
            check(L.gtg_adam_step(None, None, None, None, 0, ptr(a["adam"]), lr, b1, b2, 1e-8, 1.0, st), "adam_step")
        before = L.gtg_launch_count()
        check(L.gtg_head_fwd_bwd(ptr(h), ptr(Wmu), ptr(bmu), ptr(Wlv), ptr(blv), ptr(D1), ptr(c1), ptr(D2), ptr(c2),
                                 ptr(eps), ptr(target), B, N, d, hid, ptr(klw), klw[1:].data_ptr(), ptr(b["mu"]),
                                 ptr(b["lv"]), ptr(b["z"]), ptr(b["y1"]), ptr(b["out"]), ptr(b["dl"]), ptr(b["sc"]),
                                 b["sc"][1:].data_ptr(), ptr(b["dW"][:d]), ptr(b["dbmu"]), ptr(b["dW"][d:]), ptr(b["dblv"]),
                                 ptr(b["dD1"]), ptr(b["dc1"]), ptr(b["dD2"]), ptr(b["dc2"]), ptr(b["dz"]), ptr(b["dh"]),
                                 ptr(b["adam"]) if adam else None, lr, b1, b2, ptr(ws), nb, st), "head")
        launches = L.gtg_launch_count() - before
        torch.cuda.synchronize()
        for k in a:
            assert not torch.isnan(b[k]).any(), k
            if k == "sc":
                # the KL term is the same sum; the reconstruction loss is summed from per-wave partials of the
                # decoder's last product instead of per-graph ones: same terms, another order of additions
                assert a[k][1].item() == b[k][1].item()
                assert b[k][0].item() == pytest.approx(a[k][0].item(), rel=2e-6)
                continue
            assert torch.equal(a[k], b[k]), (rep, k, (a[k] - b[k]).abs().max().item())
        if 256 <= B <= 1024 and d % 64 == 0 and hid % 64 == 0:
            # batch sizes whose weight-gradient launches carry the side jobs, widths whose products all pair up
            assert launches <= 8, launches


@pytest.mark.parametrize("B,N,d,hid", CONFIGS)
def test_module_step_matches_oracle(B, N, d, hid):
    """GTVAE(text=False).forward/backward through autograd == oracle."""
    from gtg_vae_amd import GTVAE
    from gtg_vae_amd.data import synth_graph_batch
    from oracle import graphvae_oracle as O

    batch = synth_graph_batch(B, N, seed=21)
    params = _oracle_params(N, d, hid, seed=4)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    eps = torch.randn(B, d, generator=torch.Generator().manual_seed(8))
    kl_w = 0.3
    loss, rec, kl, out = O.graph_step_loss(p, None, batch.edge_index, batch.adj_padded, eps, kl_w, B, N)
    loss.backward()

    model = GTVAE(N, d, hid, N, text=False).cuda()
    sd = {("graph_vae." + k if not k.startswith("zg_") else k): v for k, v in params.items()}
    model.load_state_dict(sd)
    gb = batch.to("cuda")
    res = model(gb.dense_x(), gb.edge_index, gb.batch, gb.adj_padded, gb.gold_edges, None, kl_w, eps=eps.cuda())
    gloss, _, gkl, grec = res[0], res[1], res[2], res[3]
    assert abs(gloss.item() - loss.item()) <= 1e-5 * abs(loss.item())
    assert abs(grec.item() - rec.item()) <= 1e-5 * abs(rec.item())
    assert abs(gkl.item() - kl.item()) <= 1e-5 * abs(kl.item())
    gloss.backward()
    got = dict(model.named_parameters())
    for k, v in p.items():
        key = "graph_vae." + k if not k.startswith("zg_") else k
        assert _relerr(got[key].grad, v.grad) < 1e-4, k


@pytest.mark.parametrize("B,N,d,hid", CONFIGS)
def test_trainer_step_and_adam_match_oracle(B, N, d, hid):
    from gtg_vae_amd.data import synth_graph_batch
    from gtg_vae_amd.trainer import GraphVAETrainer
    from oracle import graphvae_oracle as O

    params = _oracle_params(N, d, hid, seed=6)
    tr = GraphVAETrainer(N, d, hid, N, device="cuda", init=params)
    p = {k: v.clone().requires_grad_(True) for k, v in params.items()}
    m = {k: torch.zeros_like(v) for k, v in p.items()}
    v2 = {k: torch.zeros_like(v) for k, v in p.items()}
    for step in range(1, 4):
        batch = synth_graph_batch(B, N, seed=30 + step)
        eps = torch.randn(B, d, generator=torch.Generator().manual_seed(step))
        kl_w = 0.1 * step
        for t in p.values():
            t.grad = None
        loss, rec, kl, _ = O.graph_step_loss(p, None, batch.edge_index, batch.adj_padded, eps, kl_w, B, N)
        loss.backward()
        res = tr.step(batch.to("cuda"), eps.cuda(), kl_w=kl_w, update=True)
        assert tr.edges_ok()
        assert abs(res["recon"].item() - rec.item()) <= 1e-5 * abs(rec.item())
        assert abs(res["kl"].item() - kl.item()) <= 1e-5 * abs(kl.item()) + 1e-7
        for k in p:
            assert _relerr(tr.named_grads()[k], p[k].grad) < 1e-4, (step, k)
        with torch.no_grad():
            for k in p:
                O.adam_step(p[k], p[k].grad, m[k], v2[k], step)
        for k in p:
            # Adam normalises the step to ~lr per element, so compare parameters absolutely
            assert (tr.named_params()[k].cpu() - p[k].detach()).abs().max().item() < 2e-5, (step, k)


def test_step_is_bitwise_deterministic_and_graph_replay_matches():
    from gtg_vae_amd.data import synth_graph_batch
    from gtg_vae_amd.trainer import GraphVAETrainer

    # the second shape is large enough that every buffer clear inside the step spans many blocks
    for B, N, d, hid in ((32, 32, 64, 128), (256, 64, 64, 64)):
        params = _oracle_params(N, d, hid, seed=9)
        # different batches (and edge counts) from step to step: a replayed graph must pick up the new inputs
        batches = [synth_graph_batch(B, N, seed=77 + i).to("cuda") for i in range(3)]
        eps = [torch.randn(B, d, generator=torch.Generator().manual_seed(5 + i)).cuda() for i in range(3)]
        max_e = max(b.edge_index.shape[1] for b in batches)
        runs = []
        for use_graph in (False, False, True):
            tr = GraphVAETrainer(N, d, hid, N, device="cuda", init=params, max_batch=B, max_edges=max_e)
            trace = []
            for i in range(5):
                res = tr.step(batches[i % 3], eps[i % 3], kl_w=0.5, update=True, use_graph=use_graph)
                trace.append((res["recon"].item(), res["kl"].item()))
                assert tr.edges_ok()
            torch.cuda.synchronize()
            runs.append((tr.flat_p.clone(), tr.flat_g.clone(), trace))
        assert torch.equal(runs[0][0], runs[1][0]) and torch.equal(runs[0][1], runs[1][1])
        assert torch.equal(runs[0][0], runs[2][0]) and torch.equal(runs[0][1], runs[2][1])
        assert runs[0][2] == runs[1][2] == runs[2][2]


def test_loss_report_precision_recall_matches_oracle():
    from gtg_vae_amd import GraphVAE
    from gtg_vae_amd.data import synth_graph_batch
    from oracle import mst_oracle

    B, N = 24, 10
    batch = synth_graph_batch(B, N, seed=3)
    model = GraphVAE(N, 32, 64, N, 1).cuda()
    out = torch.rand(B, N * (N + 1) // 2, generator=torch.Generator().manual_seed(1)).cuda()
    _, rec, prec = model.loss(out, batch.adj_padded.numpy(), batch.gold_edges, True)
    r32, p32 = np.float32(0), np.float32(0)
    for b in range(B):
        edges = mst_oracle.mst_edges_from_triu(out[b].cpu().numpy(), N)
        r, p = mst_oracle.edge_precision_recall(edges, batch.gold_edges[b])
        r32 = np.float32(r32 + np.float32(r))
        p32 = np.float32(p32 + np.float32(p))
    assert rec.item() == float(np.float32(r32) / np.float32(B))
    assert prec.item() == float(np.float32(p32) / np.float32(B))
    # report=False returns exact zeros like the reference
    _, r0, p0 = model.loss(out, batch.adj_padded.numpy(), batch.gold_edges, False)
    assert r0.item() == 0.0 and p0.item() == 0.0


def test_trainer_reports_a_bad_edge_once_and_recovers():
    """An edge that leaves its graph sets the (sticky) device flag; edges_ok() reports it once, and the next
    good batch through the same buffers is clean again -- also a check that the count scratch is left zero."""
    from gtg_vae_amd.data import synth_graph_batch
    from gtg_vae_amd.trainer import GraphVAETrainer
    from oracle import graphvae_oracle as O

    B, N, d, hid = 4, 10, 16, 32
    params = O.init_params(N, d, hid, N, seed=0)
    tr = GraphVAETrainer(N, d, hid, N, device="cuda", init=params)
    good = synth_graph_batch(B, N, seed=5)
    eps = torch.randn(B, d, generator=torch.Generator().manual_seed(1))
    want = tr.step(good.to("cuda"), eps.cuda(), kl_w=0.5, update=False)["recon"].item()
    assert tr.edges_ok()
    bad = synth_graph_batch(B, N, seed=5)
    bad.edge_index = torch.cat([bad.edge_index, torch.tensor([[1], [N + 3]])], 1)  # graph 0 -> graph 1
    tr.step(bad.to("cuda"), eps.cuda(), kl_w=0.5, update=False)
    assert not tr.edges_ok()
    got = tr.step(good.to("cuda"), eps.cuda(), kl_w=0.5, update=False)["recon"].item()
    assert tr.edges_ok() and got == want


def test_cpu_tensors_fail_loudly():
    from gtg_vae_amd import functional as Fn
    from gtg_vae_amd._lib import GtgError

    with pytest.raises(GtgError):
        Fn.build_adj(torch.zeros((2, 1), dtype=torch.long), 1, 4)
