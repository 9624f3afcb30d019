"""The fp32 MFMA GEMM against float64 matmul: the three operand layouts, edge
tiles, unaligned leading dimensions, split-K and every epilogue."""
import pytest
import torch

pytestmark = pytest.mark.gpu

SHAPES = [(128, 128, 32), (256, 384, 96), (129, 67, 35), (10, 55, 10), (640, 64, 10), (512, 2080, 512),
          (2080, 512, 512), (64, 1024, 4096), (1000, 300, 77), (4096, 512, 1024)]


def _ref(A, B):
    return (A.double() @ B.double())


def _tol(K):
    return 2e-6 * max(K, 1) ** 0.5 + 1e-6


@pytest.mark.parametrize("M,N,K", SHAPES)
@pytest.mark.parametrize("layout", ["NT", "NN", "TN"])
def test_layouts(M, N, K, layout):
    from gtg_vae_amd import functional as Fn

    torch.manual_seed(M * 7 + N * 3 + K)
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(K, N, device="cuda")
    want = _ref(A, B)
    if layout == "NT":
        got = Fn.gemm(A, B.t().contiguous(), True, True)
    elif layout == "NN":
        got = Fn.gemm(A, B, True, False)
    else:
        got = Fn.gemm(A.t().contiguous(), B, False, False)
    err = (got.double() - want).abs().max().item()
    scale = want.abs().max().item()
    assert err <= _tol(K) * max(scale, 1.0) * 4, (layout, M, N, K, err)


def test_asymmetric_identity_catches_transposed_output():
    from gtg_vae_amd import functional as Fn

    n = 96
    I = torch.eye(n, device="cuda")
    B = torch.arange(n * 70, device="cuda", dtype=torch.float32).reshape(n, 70) % 251
    assert torch.equal(Fn.gemm(I, B, True, False), B)
    assert torch.equal(Fn.gemm(I, B.t().contiguous(), True, True), B)
    assert torch.equal(Fn.gemm(I, B, False, False), B)


def test_epilogues():
    from gtg_vae_amd import functional as Fn

    torch.manual_seed(0)
    M, N, K = 200, 150, 64
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda")
    bias = torch.randn(N, device="cuda")
    y = (A.double() @ W.double().t() + bias.double())
    assert torch.allclose(Fn.gemm(A, W, bias=bias, epilogue=0).double(), y, atol=1e-4)
    assert torch.allclose(Fn.gemm(A, W, bias=bias, epilogue=1).double(), y.clamp(min=0), atol=1e-4)
    assert torch.allclose(Fn.gemm(A, W, bias=bias, epilogue=2).double(), torch.sigmoid(y), atol=1e-5)
    aux = torch.randn(M, N, device="cuda")
    want = (A.double() @ W.double().t()) * (aux > 0)
    assert torch.allclose(Fn.gemm(A, W, epilogue=3, aux=aux).double(), want, atol=1e-4)
    C = torch.randn(M, N, device="cuda")
    want = C.double() + A.double() @ W.double().t()
    assert torch.allclose(Fn.gemm(A, W, epilogue=4, out=C.clone()).double(), want, atol=1e-4)


def test_split_k_is_deterministic_and_correct():
    from gtg_vae_amd import functional as Fn

    torch.manual_seed(1)
    K, M, N = 32768, 512, 1024
    A = torch.randn(K, M, device="cuda")
    B = torch.randn(K, N, device="cuda")
    got = Fn.gemm(A, B, False, False)
    got2 = Fn.gemm(A, B, False, False)
    assert torch.equal(got, got2)
    want = A.double().t() @ B.double()
    assert (got.double() - want).abs().max().item() < 2e-3
    nosplit = Fn.gemm(A, B, False, False, split_k=False)
    assert (nosplit.double() - want).abs().max().item() < 5e-2  # one sequential fp32 chain over K=32768
