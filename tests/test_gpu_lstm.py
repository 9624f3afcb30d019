"""The HIP LSTM recurrence (gtg_lstm_fwd / gtg_lstm_bwd through functional.lstm) against the CPU oracle's
explicit cell recurrence (oracle/textvae_oracle.py: lstm, pinned to the reference's own nn.LSTM outputs by
tests/test_oracle_textvae.py), forward and backward, both directions, with and without an initial state, at
small ragged sizes and at the text branch's size (text_vae.py:37-40: nn.LSTM(512, 1024), batch 256, 40 steps).
Tolerances: outputs 1e-5, gradients 1e-4, relative to the largest entry of the tensor compared."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _relerr(a, b):
    a, b = a.detach().double().cpu(), b.detach().double().cpu()
    return (a - b).abs().max().item() / (b.abs().max().item() + 1e-30)


def _params(I, H, seed, scale):
    g = torch.Generator().manual_seed(seed)
    mk = lambda *s: (torch.rand(*s, generator=g) * 2 - 1) * scale
    return mk(4 * H, I), mk(4 * H, H), mk(4 * H), mk(4 * H)


CASES = [
    # T, B, I, H, reverse, with initial state
    (1, 3, 8, 64, False, False),
    (5, 6, 16, 64, False, True),
    (5, 6, 16, 64, True, True),
    (9, 70, 36, 128, True, False),
    (21, 33, 512, 1024, False, False),
    (40, 256, 512, 1024, True, False),
]


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2", "f32"])
@pytest.mark.parametrize("T,B,I,H,reverse,init", CASES)
def test_lstm_forward_and_backward_match_oracle(T, B, I, H, reverse, init, mode):
    from gtg_vae_amd import functional as Fn
    from oracle import textvae_oracle as TO

    g = torch.Generator().manual_seed(T * 1000 + B)
    scale = 1.0 / H ** 0.5
    w_ih, w_hh, b_ih, b_hh = _params(I, H, 7, scale)
    x = torch.randn(T, B, I, generator=g)
    h0 = torch.randn(B, H, generator=g) * 0.5 if init else None
    c0 = torch.randn(B, H, generator=g) * 0.5 if init else None
    # upstream gradients of all three results
    d_hs = torch.randn(T, B, H, generator=g)
    d_hT = torch.randn(B, H, generator=g)
    d_cT = torch.randn(B, H, generator=g)

    leaves = [t.clone().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
    init_leaves = [t.clone().requires_grad_(True) for t in (h0, c0)] if init else [None, None]
    out, hT, cT = TO.lstm(leaves[0], leaves[1], leaves[2], leaves[3], leaves[4], init_leaves[0], init_leaves[1], reverse)
    ((out * d_hs).sum() + (hT * d_hT).sum() + (cT * d_cT).sum()).backward()

    cu = [t.clone().cuda().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
    cu_init = [t.clone().cuda().requires_grad_(True) for t in (h0, c0)] if init else [None, None]
    hs, ghT, gcT = Fn.lstm(cu[0], cu[1], cu[2], cu[3], cu[4], cu_init[0], cu_init[1], reverse, mode)
    assert _relerr(hs, out) < 1e-5 and _relerr(ghT, hT) < 1e-5 and _relerr(gcT, cT) < 1e-5
    ((hs * d_hs.cuda()).sum() + (ghT * d_hT.cuda()).sum() + (gcT * d_cT.cuda()).sum()).backward()
    names = ("x", "w_ih", "w_hh", "b_ih", "b_hh")
    for name, a, b in zip(names, cu, leaves):
        assert _relerr(a.grad, b.grad) < 1e-4, (name, _relerr(a.grad, b.grad))
    if init:
        for name, a, b in zip(("h0", "c0"), cu_init, init_leaves):
            assert _relerr(a.grad, b.grad) < 1e-4, (name, _relerr(a.grad, b.grad))


def test_lstm_random_shapes_match_oracle():
    """Sixteen random shapes (hidden sizes 64..320, inputs 4..132, batches 1..150 -- ragged against the 64-row
    tiles --, 1..14 steps, either direction, any subset of the optional inputs and upstream gradients)."""
    import random

    from gtg_vae_amd import functional as Fn
    from oracle import textvae_oracle as TO

    rnd = random.Random(20261111)
    for case in range(16):
        H = 64 * rnd.randint(1, 5)
        I = 4 * rnd.randint(1, 33)
        B = rnd.choice([1, 2, 5, 63, 64, 65, rnd.randint(1, 150)])
        T = rnd.randint(1, 14)
        reverse, init = rnd.random() < 0.5, rnd.random() < 0.5
        use = [rnd.random() < 0.7 for _ in range(3)]   # which of (d_hs, d_hT, d_cT) carry a gradient
        if not any(use):
            use[rnd.randrange(3)] = True
        g = torch.Generator().manual_seed(case)
        w_ih, w_hh, b_ih, b_hh = _params(I, H, 100 + case, 1.5 / H ** 0.5)
        x = torch.randn(T, B, I, generator=g)
        h0 = torch.randn(B, H, generator=g) * 0.5 if init else None
        c0 = torch.randn(B, H, generator=g) * 0.5 if init else None
        ups = [torch.randn(T, B, H, generator=g), torch.randn(B, H, generator=g), torch.randn(B, H, generator=g)]

        def run(fn, tensors, inits, dev):
            res = fn(*tensors, inits[0], inits[1], reverse)
            return res, sum((r * u.to(dev)).sum() for r, u, on in zip(res, ups, use) if on)

        leaves = [t.clone().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
        li = [t.clone().requires_grad_(True) for t in (h0, c0)] if init else [None, None]
        want, loss = run(TO.lstm, leaves, li, "cpu")
        loss.backward()
        cu = [t.clone().cuda().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
        ci = [t.clone().cuda().requires_grad_(True) for t in (h0, c0)] if init else [None, None]
        got, gloss = run(Fn.lstm, cu, ci, "cuda")
        gloss.backward()
        tag = (case, T, B, I, H, reverse, init, use)
        for a, b in zip(got, want):
            assert _relerr(a, b) < 1e-5, tag
        for a, b in zip(cu + (ci if init else []), leaves + (li if init else [])):
            assert _relerr(a.grad, b.grad) < 1e-4, (tag, _relerr(a.grad, b.grad))


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_lstm_initial_state_of_any_magnitude(mode):
    """An initial hidden state is not bounded by 1 like the states the recurrence produces (the two-plane fp16
    mode scales each of its rows by that row's largest magnitude): rows of very different size, one all zero."""
    from gtg_vae_amd import functional as Fn
    from oracle import textvae_oracle as TO

    T, B, I, H = 4, 9, 16, 128
    g = torch.Generator().manual_seed(17)
    w_ih, w_hh, b_ih, b_hh = _params(I, H, 19, 1.0 / H ** 0.5)
    x = torch.randn(T, B, I, generator=g)
    h0 = torch.randn(B, H, generator=g) * torch.tensor([1e-6, 1e-3, 0.0, 1.0, 40.0, 3e3, 0.5, 1e-9, 7.0]).view(B, 1)
    c0 = torch.randn(B, H, generator=g)
    d_hs = torch.randn(T, B, H, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh, h0, c0)]
    out, _, _ = TO.lstm(*leaves[:5], leaves[5], leaves[6], False)
    (out * d_hs).sum().backward()
    cu = [t.clone().cuda().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh, h0, c0)]
    hs, _, _ = Fn.lstm(*cu[:5], cu[5], cu[6], False, mode)
    (hs * d_hs.cuda()).sum().backward()
    assert _relerr(hs, out) < 1e-5
    for name, a, b in zip(("x", "w_ih", "w_hh", "b_ih", "b_hh", "h0", "c0"), cu, leaves):
        assert _relerr(a.grad, b.grad) < 1e-4, (name, _relerr(a.grad, b.grad))


def test_lstm_final_state_only_gradient():
    """The encoder's use (text_vae.py:57-58): only h_n feeds the loss, the output sequence gets no gradient."""
    from gtg_vae_amd import functional as Fn
    from oracle import textvae_oracle as TO

    T, B, I, H = 7, 10, 32, 64
    g = torch.Generator().manual_seed(3)
    w_ih, w_hh, b_ih, b_hh = _params(I, H, 11, 0.2)
    x = torch.randn(T, B, I, generator=g)
    d_hT = torch.randn(B, H, generator=g)
    for reverse in (False, True):
        leaves = [t.clone().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
        _, hT, _ = TO.lstm(*leaves, reverse=reverse)
        (hT * d_hT).sum().backward()
        cu = [t.clone().cuda().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
        _, ghT, _ = Fn.lstm(*cu, reverse=reverse)
        (ghT * d_hT.cuda()).sum().backward()
        assert _relerr(ghT, hT) < 1e-5
        for a, b in zip(cu, leaves):
            assert _relerr(a.grad, b.grad) < 1e-4


def test_module_routes_through_the_hip_recurrence_and_matches_stock_lstm():
    """DAE.run_lstm: same results as the stock nn.LSTM module it wraps (both directions, given hidden state),
    in the default matrix mode and in the fp32 one (the library's recurrence runs in both: there is no vendor-LSTM
    path on the GPU), and sizes the recurrence does not take are an error."""
    from gtg_vae_amd.text_vae import DAE

    torch.manual_seed(0)
    m = DAE(32, 50, 0).cuda().train()
    emb = torch.randn(6, 5, 512, device="cuda")
    hidden = (torch.randn(1, 5, 1024, device="cuda") * 0.3, torch.randn(1, 5, 1024, device="cuda") * 0.3)
    for mod, hid in ((m.E, None), (m.G, None), (m.G, hidden)):
        want_out, (want_h, want_c) = mod(emb, hid)
        out, (h, c) = m.run_lstm(mod, emb, hid)
        assert out.shape == want_out.shape and h.shape == want_h.shape and c.shape == want_c.shape
        assert _relerr(out, want_out) < 1e-5 and _relerr(h, want_h) < 1e-5 and _relerr(c, want_c) < 1e-5
    m.matrix_mode = "f32"
    out, (h, _) = m.run_lstm(m.G, emb, None)
    want_out, (want_h, _) = m.G(emb)
    assert _relerr(out, want_out) < 1e-5 and _relerr(h, want_h) < 1e-5
    assert not torch.equal(out, want_out)          # not the stock module's kernels
    from gtg_vae_amd import _lib

    odd = torch.nn.LSTM(512, 96, 1).cuda()          # hidden size not a multiple of 64
    with pytest.raises(_lib.GtgError):
        m.run_lstm(odd, emb, None)


def test_lstm_bf16_mode_stays_close():
    """matrix mode "bf16": the products over all steps (input projection, dW, dx) run on operands rounded to
    bf16, the recurrence itself stays on three planes.  Tolerance of that mode: outputs 2e-2 of the largest
    entry, gradients cosine 0.999."""
    from gtg_vae_amd import functional as Fn
    from oracle import textvae_oracle as TO

    T, B, I, H = 12, 40, 64, 128
    g = torch.Generator().manual_seed(5)
    w_ih, w_hh, b_ih, b_hh = _params(I, H, 13, 1.0 / H ** 0.5)
    x = torch.randn(T, B, I, generator=g)
    d_hs = torch.randn(T, B, H, generator=g)
    leaves = [t.clone().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
    out, _, _ = TO.lstm(*leaves)
    (out * d_hs).sum().backward()
    cu = [t.clone().cuda().requires_grad_(True) for t in (x, w_ih, w_hh, b_ih, b_hh)]
    hs, _, _ = Fn.lstm(*cu, matrix_mode="bf16")
    (hs * d_hs.cuda()).sum().backward()
    assert _relerr(hs, out) < 2e-2
    for a, b in zip(cu, leaves):
        cos = torch.nn.functional.cosine_similarity(a.grad.detach().cpu().flatten().double(), b.grad.flatten().double(), dim=0)
        assert cos > 0.999, float(cos)


def test_lstm_rejects_sizes_it_does_not_take():
    from gtg_vae_amd import _lib
    from gtg_vae_amd import functional as Fn

    w_ih, w_hh, b_ih, b_hh = (t.cuda() for t in _params(8, 48, 1, 0.1))
    with pytest.raises(_lib.GtgError):
        Fn.lstm(torch.randn(3, 2, 8, device="cuda"), w_ih, w_hh, b_ih, b_hh)
