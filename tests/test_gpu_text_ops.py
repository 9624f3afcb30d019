"""GPU tests of the text branch's gather / scatter / noise kernels (csrc/text_ops.hip) against the CPU oracle and
the reference's own word_drop outputs: word drop (noise.py:12-22), embedding lookup + dropout + added latent row
(text_vae.py:56, 62) with its ordered scatter-add backward, dropout (text_vae.py:64)."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "noise_golden.npz")


@pytest.mark.parametrize("tag", ["small", "batch", "long", "keep_all", "drop_most"])
def test_word_drop_with_the_references_keep_flags_is_bit_exact(tag):
    """tokens + the keep flags the reference drew -> exactly the reference's output (integer work: bit-exact)."""
    from gtg_vae_amd import functional as Fn

    g = np.load(GOLD)
    pad = int(g[tag + "__meta"][0])
    x = torch.from_numpy(g[tag + "__x"]).cuda()
    keep = torch.from_numpy(g[tag + "__keep"]).cuda()
    got = Fn.word_drop(x, pad, 0.3, keep)
    assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), g[tag + "__out"])


def test_word_drop_random_mode_keeps_the_start_symbol_order_and_rate():
    """With the kernel's own generator: the first token of every sentence survives, every output column is a
    subsequence of its input followed by padding only, the kept fraction is 1 - p within 5 sigma, the same
    torch seed repeats the draw and a mask drawn from a torch generator equals the oracle on that mask."""
    from gtg_vae_amd import functional as Fn
    from gtg_vae_amd.noise import word_drop
    from oracle import textvae_oracle as TO

    T, B, pad, p = 40, 256, 0, 0.3
    x = torch.randint(5, 30000, (T, B))
    xc = x.cuda()
    torch.manual_seed(7)
    a = Fn.word_drop(xc, pad, p).cpu()
    torch.manual_seed(7)
    b = Fn.word_drop(xc, pad, p).cpu()
    assert torch.equal(a, b)
    c = Fn.word_drop(xc, pad, p).cpu()
    assert not torch.equal(a, c)                       # a fresh stream id per call
    assert torch.equal(a[0], x[0])
    kept = 0
    for col in range(B):
        out = a[:, col].tolist()
        n = len(out)
        while n > 0 and out[n - 1] == pad:
            n -= 1
        kept += n
        it = iter(x[:, col].tolist())
        assert all(tok in it for tok in out[:n])       # ordered subsequence
    expect = B + (T - 1) * B * (1 - p)
    sigma = ((T - 1) * B * p * (1 - p)) ** 0.5
    assert abs(kept - expect) < 5 * sigma
    gen = torch.Generator().manual_seed(3)
    got = word_drop(pad, xc, p, gen)
    keep = torch.rand((T, B), generator=torch.Generator().manual_seed(3)) > p
    assert torch.equal(got.cpu(), TO.word_drop(x, keep, pad))


def _ids(T, B, V, seed, heavy=True):
    g = torch.Generator().manual_seed(seed)
    ids = torch.randint(0, V, (T, B), generator=g)
    if heavy:
        ids[0] = 1                                     # every sentence starts with the same symbol
        for col in range(B):                           # and ends in a run of padding (id 0)
            n = int(torch.randint(2, T + 1, (1,), generator=g))
            ids[n:, col] = 0
    return ids


@pytest.mark.parametrize("T,B,V,D,with_add", [(3, 2, 7, 4, False), (40, 256, 30522, 512, True), (9, 300, 50, 36, True),
                                            (64, 70, 1000, 512, False)])
def test_embed_lookup_and_scatter_add_match_oracle(T, B, V, D, with_add):
    """p = 0: forward exact (a gather and an add), backward against a float64 index_add (heavy duplicates: the pad
    id holds thousands of rows, the start symbol a row per sentence), the added row's gradient against the sum
    over time; two runs bit-identical."""
    from gtg_vae_amd import functional as Fn
    from oracle import textvae_oracle as TO

    g = torch.Generator().manual_seed(T * 7 + B)
    table = torch.randn(V, D, generator=g)
    add = torch.randn(B, D, generator=g) if with_add else None
    ids = _ids(T, B, V, 5)
    dout = torch.randn(T, B, D, generator=g)

    w = table.clone().cuda().requires_grad_(True)
    a = add.clone().cuda().requires_grad_(True) if with_add else None
    out = Fn.embed_dropout(ids.cuda(), w, 0.0, a)
    want = TO.embed_dropout(table, ids, None, 0.0, add)
    assert torch.equal(out.cpu(), want)
    out.backward(dout.cuda())
    ref = torch.zeros(V, D, dtype=torch.float64).index_add_(0, ids.reshape(-1), dout.reshape(-1, D).double())
    err = (w.grad.double().cpu() - ref).abs().max().item() / (ref.abs().max().item() + 1e-30)
    assert err < 2e-6, err
    if with_add:
        ref_add = dout.double().sum(0)
        assert (a.grad.double().cpu() - ref_add).abs().max().item() / ref_add.abs().max().item() < 2e-6
    w2 = table.clone().cuda().requires_grad_(True)
    Fn.embed_dropout(ids.cuda(), w2, 0.0, None).backward(dout.cuda())
    assert torch.equal(w.grad, w2.grad)                # ordered sums: bit-reproducible


def test_embed_gradient_accumulates_in_place_into_a_flat_buffer_view():
    """A weight marked ``grad_inplace`` whose .grad exists (JointTrainer's flat buffer) gets the scatter-add on
    top of what is there and autograd receives no tensor for it; two uses of the table in one graph add up."""
    from gtg_vae_amd import functional as Fn

    T, B, V, D = 12, 9, 40, 8
    g = torch.Generator().manual_seed(1)
    table = torch.randn(V, D, generator=g)
    ids1, ids2 = _ids(T, B, V, 2), _ids(T, B, V, 3)
    d1, d2 = torch.randn(T, B, D, generator=g), torch.randn(T, B, D, generator=g)
    w = torch.nn.Parameter(table.clone().cuda())
    start = torch.randn(V, D, generator=g)
    w.grad = start.clone().cuda()
    w.grad_inplace = True
    keep_ptr = w.grad.data_ptr()
    ((Fn.embed_dropout(ids1.cuda(), w) * d1.cuda()).sum() + (Fn.embed_dropout(ids2.cuda(), w) * d2.cuda()).sum()).backward()
    ref = start.double().index_add_(0, ids1.reshape(-1), d1.reshape(-1, D).double()) \
        .index_add_(0, ids2.reshape(-1), d2.reshape(-1, D).double())
    assert w.grad.data_ptr() == keep_ptr
    assert (w.grad.double().cpu() - ref).abs().max().item() < 1e-5


def test_embed_dropout_mask_is_the_same_in_forward_and_backward():
    """p = 0.5: every output element is 0 or twice the table entry, about half are kept, and the backward of
    an all-ones gradient equals the scatter-add of exactly the forward's keep pattern times 2."""
    from gtg_vae_amd import functional as Fn

    T, B, V, D, p = 40, 64, 500, 512, 0.5
    g = torch.Generator().manual_seed(4)
    table = torch.rand(V, D, generator=g) + 0.5        # no zeros: a zero output is a dropped element
    ids = _ids(T, B, V, 6)
    w = table.clone().cuda().requires_grad_(True)
    torch.manual_seed(11)
    out = Fn.embed_dropout(ids.cuda(), w, p)
    o = out.detach().cpu()
    keep = o != 0
    assert torch.equal(torch.where(keep, table[ids] * 2.0, torch.zeros(())), o)
    frac = keep.float().mean().item()
    assert abs(frac - 0.5) < 5 * (0.25 / keep.numel()) ** 0.5
    out.backward(torch.ones_like(out))
    ref = torch.zeros(V, D, dtype=torch.float64).index_add_(0, ids.reshape(-1), keep.reshape(-1, D).double() * 2.0)
    assert torch.equal(w.grad.double().cpu(), ref)     # sums of small integers: exact
    torch.manual_seed(11)
    again = Fn.embed_dropout(ids.cuda(), table.clone().cuda(), p)
    assert torch.equal(again.cpu(), o)                 # the same torch seed repeats the mask


def test_dropout_forward_and_backward_share_the_mask():
    from gtg_vae_amd import functional as Fn

    x = (torch.rand(40, 256, 1024) + 0.5).cuda().requires_grad_(True)
    torch.manual_seed(5)
    y = Fn.dropout(x, 0.5, training=True)
    keep = y.detach() != 0
    assert torch.equal(torch.where(keep, x.detach() * 2.0, torch.zeros((), device="cuda")), y.detach())
    assert abs(keep.float().mean().item() - 0.5) < 5 * (0.25 / keep.numel()) ** 0.5
    y.backward(torch.ones_like(y))
    assert torch.equal(x.grad, keep.float() * 2.0)
    assert Fn.dropout(x, 0.5, training=False) is x and Fn.dropout(x, 0.0) is x


def test_embed_flags_an_id_outside_the_table():
    from gtg_vae_amd import _lib
    from gtg_vae_amd import functional as Fn

    w = torch.randn(10, 8, device="cuda")
    Fn.EmbedDropoutFn.check_ids = True
    try:
        with pytest.raises(_lib.GtgError):
            Fn.embed_dropout(torch.tensor([[1, 10]], device="cuda"), w)
        assert Fn.embed_dropout(torch.tensor([[1, 9]], device="cuda"), w).shape == (1, 2, 8)
    finally:
        Fn.EmbedDropoutFn.check_ids = False


@pytest.mark.parametrize("lead,I,O,bias", [((256,), 2048, 256, True), ((256,), 256, 512, True), ((3, 5), 12, 7, False),
                                          ((1,), 4, 4, True)])
def test_linear_on_the_library_gemm_matches_float64(lead, I, O, bias):
    """functional.linear (hgen, z2emb of text_vae.py:41-42): forward, dx, dW and db against float64, fp32
    tolerance 2e-6 of the largest entry (products of 4 .. 2048 terms on the fp32 matrix pipe)."""
    from gtg_vae_amd import functional as Fn

    g = torch.Generator().manual_seed(I + O)
    x = torch.randn(*lead, I, generator=g)
    W = torch.randn(O, I, generator=g) / I ** 0.5
    b = torch.randn(O, generator=g) if bias else None
    dy = torch.randn(*lead, O, generator=g)
    xc, Wc = x.cuda().requires_grad_(True), W.cuda().requires_grad_(True)
    bc = b.cuda().requires_grad_(True) if bias else None
    y = Fn.linear(xc, Wc, bc)
    assert y.shape == (*lead, O)
    y.backward(dy.cuda())
    x64, W64 = x.double().requires_grad_(True), W.double().requires_grad_(True)
    b64 = b.double().requires_grad_(True) if bias else None
    ref = torch.nn.functional.linear(x64, W64, b64)
    ref.backward(dy.double())
    pairs = [(y, ref), (xc.grad, x64.grad), (Wc.grad, W64.grad)] + ([(bc.grad, b64.grad)] if bias else [])
    for got, want in pairs:
        err = (got.detach().double().cpu() - want.detach()).abs().max().item() / want.detach().abs().max().item()
        assert err < 2e-6, err


def test_weight_gradients_stored_into_their_flat_buffer_views_equal_the_returned_ones():
    """Weights marked ``grad_overwrite`` whose .grad exists (JointTrainer's flat buffer): the vocabulary projection
    and the LSTM store dW / dW_ih / dW_hh there -- bit-identical to the tensors they return to autograd otherwise,
    whatever the slot held before --, autograd receives nothing for them while their post-accumulate hooks still
    run once each (data-parallel bucket counting relies on that), and the other gradients are unchanged."""
    from gtg_vae_amd import functional as Fn

    g = torch.Generator().manual_seed(9)
    R, K, V, T, B, I, H = 70, 64, 300, 5, 14, 16, 64
    Hm = torch.randn(R, K, generator=g).cuda()
    tg = torch.randint(0, V, (R,), generator=g).cuda()
    tg[::7] = 0
    x = torch.randn(T, B, I, generator=g).cuda()
    dhs = torch.randn(T, B, H, generator=g).cuda()

    def params():
        gg = torch.Generator().manual_seed(10)
        mk = lambda *s: torch.nn.Parameter((torch.randn(*s, generator=gg) * 0.1).cuda())
        return mk(V, K), mk(V), mk(4 * H, I), mk(4 * H, H), mk(4 * H), mk(4 * H)

    def run(ps):
        W, b, w_ih, w_hh, b_ih, b_hh = ps
        loss, _ = Fn.proj_ce(Hm, W, b, tg, 0, "f16x2")
        hs, hn, cn = Fn.lstm(x, w_ih, w_hh, b_ih, b_hh, None, None, False, "f16x2")
        (loss.sum() + (hs * dhs).sum() + hn.sum()).backward()

    plain = params()
    run(plain)
    stored = params()
    calls = []
    for i in (0, 2, 3):
        stored[i].grad = torch.full_like(stored[i], 123.0)      # stale contents: must be replaced, not added to
        stored[i].grad_overwrite = True
        stored[i].register_post_accumulate_grad_hook(lambda p, i=i: calls.append(i))
    slots = [stored[i].grad.data_ptr() for i in (0, 2, 3)]
    run(stored)
    assert sorted(calls) == [0, 2, 3]
    assert [stored[i].grad.data_ptr() for i in (0, 2, 3)] == slots
    for p, q in zip(plain, stored):
        assert torch.equal(p.grad, q.grad)
