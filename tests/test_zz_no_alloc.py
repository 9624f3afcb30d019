"""The allocation-free form of the reconstruction loss, gtg_bce_loss_fwd_noalloc, against gtg_bce_loss_fwd
(which carves its partial sums from a caller workspace): both called through the C ABI on the same device
buffers, twice in a row, the second time on a smaller batch.  Same kernels, so loss and dlogits must be
bit-identical; the loss is also checked against the CPU oracle."""
import ctypes

import pytest
import torch

pytestmark = pytest.mark.gpu


def _both(L, out, adj, B, n, ws):
    from gtg_vae_amd._lib import check, ptr

    m = n * (n + 1) // 2
    stream = torch.cuda.current_stream().cuda_stream
    res = []
    for noalloc in (False, True):
        loss = torch.full((1,), float("nan"), device="cuda")
        dlog = torch.full((B, m), float("nan"), device="cuda")
        if noalloc:
            before = torch.cuda.memory_allocated()
            check(L.gtg_bce_loss_fwd_noalloc(ptr(out), ptr(adj), B, n, ptr(loss), ptr(dlog), None, stream),
                  "gtg_bce_loss_fwd_noalloc")
            assert torch.cuda.memory_allocated() == before
        else:
            check(L.gtg_bce_loss_fwd(ptr(out), ptr(adj), B, n, ptr(loss), ptr(dlog), None, ptr(ws), ws.numel(), stream),
                  "gtg_bce_loss_fwd")
        res.append((loss, dlog))
    torch.cuda.synchronize()
    return res


def test_no_alloc_bce_equals_workspace_bce_on_two_calls_in_a_row():
    from gtg_vae_amd import _lib
    from oracle import graphvae_oracle as O

    L = _lib.lib()
    assert L.gtg_bce_loss_fwd_noalloc_max_graphs() == 65536
    gen = torch.Generator().manual_seed(11)
    # second call smaller than the first: whatever the first left in the shared array must not matter
    for B, n in ((777, 64), (130, 17), (1, 1)):
        m = n * (n + 1) // 2
        out = torch.rand(B, m, generator=gen).clamp_(1e-6, 1 - 1e-6)
        out[0, 0] = 0.0          # log clamp at -100
        adj = (torch.rand(B, n, n, generator=gen) < 0.1).float()
        ws = torch.empty(L.gtg_bce_workspace_bytes(B), dtype=torch.uint8, device="cuda")
        (l0, d0), (l1, d1) = _both(L, out.cuda(), adj.cuda(), B, n, ws)
        assert torch.equal(l0, l1) and torch.equal(d0, d1), (B, n)
        want = O.recon_loss(out, adj)
        assert abs(l1.item() - want.item()) <= 1e-5 * abs(want.item())


def test_no_alloc_bce_rejects_batches_beyond_the_static_scratch():
    from gtg_vae_amd import _lib

    L = _lib.lib()
    x = torch.zeros(4, device="cuda")
    rc = L.gtg_bce_loss_fwd_noalloc(x.data_ptr(), x.data_ptr(), 65537, 1, x.data_ptr(), None, None,
                                    torch.cuda.current_stream().cuda_stream)
    assert rc == -3 and b"65536" in L.gtg_last_error()
    rc = L.gtg_bce_loss_fwd_noalloc(None, x.data_ptr(), 4, 1, x.data_ptr(), None, None, None)
    assert rc == -1
