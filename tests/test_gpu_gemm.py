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


# 128x128 tiles over 512 resident block slots: (rows, N) picked so that the last round is empty, nearly
# empty (many parts per tile), half full and nearly full (no split), with row counts known to the host
# and read from the device
@pytest.mark.parametrize("M,rows,N,K", [
    (20800, 20763, 512, 1024),   # 652 tiles: 140 in the tail, three parts each
    (20800, None, 512, 1024),
    (16384, 16384, 512, 256),    # exactly one round
    (16500, 16500, 512, 1024),   # 4 tail tiles (capped by the k steps per part)
    (33000, 24000, 1024, 512),   # tail above half the slots: no split
    (33000, 129, 1024, 512),     # a single round, partly filled
    (700, 1, 256, 2048),
])
@pytest.mark.parametrize("layout", ["NT", "NN"])
def test_rows_on_device_with_tail_split(M, rows, N, K, layout):
    from gtg_vae_amd import functional as Fn

    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda") if layout == "NT" else torch.randn(K, N, device="cuda")
    bias = torch.randn(N, device="cuda")
    rows_dev = None if rows is None else torch.tensor([rows], dtype=torch.int32, device="cuda")
    live = M if rows is None else rows
    out = torch.full((M, N), 7.0, device="cuda")
    got = Fn.gemm_rows(A, B, True, layout == "NT", rows_dev=rows_dev, bias=bias, epilogue=1, out=out).clone()
    again = Fn.gemm_rows(A, B, True, layout == "NT", rows_dev=rows_dev, bias=bias, epilogue=1, out=out)
    assert torch.equal(got, again)                       # fixed summation order: bitwise reproducible
    assert bool((got[live:] == 7.0).all())               # rows past the device count are not written
    want = torch.relu(A[:live].double() @ (B.double().t() if layout == "NT" else B.double()) + bias.double())
    err = (got[:live].double() - want).abs().max().item()
    assert err < _tol(K) * 40, err
    # the plain kernel (no tail split) on the same rows
    plain = Fn.gemm(A[:live], B, True, layout == "NT", bias=bias, epilogue=1)
    assert (plain - got[:live]).abs().max().item() < _tol(K) * 40


def test_rows_on_device_mask_epilogue():
    from gtg_vae_amd import functional as Fn

    torch.manual_seed(3)
    M, rows, N, K = 20800, 20500, 1024, 512
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(K, N, device="cuda")
    aux = torch.randn(M, N, device="cuda")
    rows_dev = torch.tensor([rows], dtype=torch.int32, device="cuda")
    got = Fn.gemm_rows(A, B, True, False, rows_dev=rows_dev, epilogue=3, aux=aux)
    want = torch.where(aux[:rows] > 0, A[:rows].double() @ B.double(), torch.zeros((), dtype=torch.float64, device="cuda"))
    assert (got[:rows].double() - want).abs().max().item() < _tol(K) * 40
    assert bool((got[rows:] == 0).all())


# ---- matrix pipes: the three-plane bf16 split and the scaled two-plane fp16 split must be as accurate as the
# fp32 pipe; plain bf16 is coarse
MODE_SHAPES = [  # all large enough for the 128x128 configuration the modes apply to
    ("NT", 8192, 512, 1024), ("NN", 4096, 1024, 512), ("TN", 2048, 2048, 4096),
    ("NT", 5000, 700, 333), ("NN", 5001, 701, 77), ("TN", 2100, 1801, 1003),
    # at least 512 tiles of 256 x 128: the eight-wave configuration of the bf16 pipes (whole and ragged tiles)
    ("NT", 8192, 2048, 256), ("NN", 4100, 4100, 130), ("TN", 4096, 4227, 203), ("NT", 16500, 1000, 48),
]


@pytest.mark.parametrize("layout,M,N,K", MODE_SHAPES)
def test_bf16x3_pipe_is_as_accurate_as_fp32_pipe(layout, M, N, K):
    from gtg_vae_amd import functional as Fn

    torch.manual_seed(M + N)
    a_kc, b_kc = layout != "TN", layout == "NT"
    A = torch.randn(M, K, device="cuda") if a_kc else torch.randn(K, M, device="cuda")
    B = torch.randn(N, K, device="cuda") if b_kc else torch.randn(K, N, device="cuda")
    # a wide dynamic range inside every row, so that the low planes of the split matter
    A *= torch.exp(3 * torch.randn_like(A))
    a, b = A.double(), B.double()
    want = (a if a_kc else a.t()) @ (b.t() if b_kc else b)
    scale = ((a.abs() if a_kc else a.abs().t()) @ (b.abs().t() if b_kc else b.abs()))  # sum |a||b| per element
    err = {}
    for mode in ("f32", "bf16x3", "f16x2", "bf16"):
        got = Fn.gemm(A, B, a_kc, b_kc, matrix_mode=mode)
        err[mode] = ((got.double() - want).abs() / scale).max().item()
    # errors are relative to sum_k |a||b|, the scale fp32 rounding acts on (the fp32 pipe itself: a few 1e-6 at K ~ 1000)
    assert err["f32"] < 2e-5, err
    assert err["bf16x3"] < 2.0 * err["f32"] + 2e-7, err
    assert err["f16x2"] < 2.0 * err["f32"] + 2e-7, err
    assert 1e-5 < err["bf16"] < 2e-2, err          # really the coarse pipe, and sane
    print("\n%s M=%d N=%d K=%d  f32 %.2e  bf16x3 %.2e  f16x2 %.2e  bf16 %.2e" % (layout, M, N, K, err["f32"], err["bf16x3"], err["f16x2"], err["bf16"]))


@pytest.mark.parametrize("mode", ["bf16x3", "f16x2"])
def test_bf16x3_pipe_epilogues_ragged_rows_and_split_k(mode):
    from gtg_vae_amd import functional as Fn

    torch.manual_seed(5)
    M, rows, N, K = 20800, 20763, 1024, 512
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(K, N, device="cuda")
    aux = torch.randn(M, N, device="cuda")
    rd = torch.tensor([rows], dtype=torch.int32, device="cuda")
    got = Fn.gemm_rows(A, B, True, False, rows_dev=rd, epilogue=3, aux=aux, matrix_mode=mode)
    want = torch.where(aux[:rows] > 0, A[:rows].double() @ B.double(), torch.zeros((), dtype=torch.float64, device="cuda"))
    assert (got[:rows].double() - want).abs().max().item() < _tol(K) * 40
    assert bool((got[rows:] == 0).all())
    # tail split (652 tiles) through the bf16 pipe, bitwise reproducible
    A2 = torch.randn(M, 1024, device="cuda")
    B2 = torch.randn(512, 1024, device="cuda")
    o1 = Fn.gemm_rows(A2, B2, True, True, rows_dev=rd, matrix_mode=mode)
    o2 = Fn.gemm_rows(A2, B2, True, True, rows_dev=rd, matrix_mode=mode)
    assert torch.equal(o1, o2)
    assert (o1[:rows].double() - A2[:rows].double() @ B2.double().t()).abs().max().item() < _tol(1024) * 40
    # split-K weight gradient
    X = torch.randn(32768, 512, device="cuda")
    Y = torch.randn(32768, 1024, device="cuda")
    g1 = Fn.gemm(X, Y, False, False, matrix_mode=mode)
    g2 = Fn.gemm(X, Y, False, False, matrix_mode=mode)
    assert torch.equal(g1, g2)
    assert (g1.double() - X.double().t() @ Y.double()).abs().max().item() < 2e-3


@pytest.mark.parametrize("sa,sb", [(1e-30, 1e20), (3e25, 1e-12), (1.0, 0.0), (7e-39, 1.0)])
def test_f16x2_pipe_rescales_tiny_and_huge_operands(sa, sb):
    """The two-plane fp16 split scales every operand by a power of two found from its largest magnitude: results
    for operands far outside fp16's range (and an all-zero one, and one of fp32 subnormals) keep the fp32-level
    error relative to sum |a||b|."""
    from gtg_vae_amd import functional as Fn

    torch.manual_seed(11)
    M, N, K = 2048, 1536, 320
    A = torch.randn(M, K, device="cuda") * sa
    B = torch.randn(N, K, device="cuda") * sb
    got = Fn.gemm(A, B, True, True, matrix_mode="f16x2")
    want = A.double() @ B.double().t()
    scale = (A.double().abs() @ B.double().abs().t()).clamp_min(1e-300)
    assert torch.isfinite(got).all()
    if sb == 0.0:
        assert float(got.abs().max()) == 0.0
    elif sa < 1e-37:   # subnormal inputs: the operand itself carries only a few bits
        assert ((got.double() - want).abs() / scale).max().item() < 1e-2
    else:
        assert ((got.double() - want).abs() / scale).max().item() < 2e-6


def test_f16x2_pipe_ignores_what_lies_past_the_device_side_extent():
    """Rows of A past the count in device memory may hold anything -- huge values, NaN: the pass that finds the
    operands' largest magnitudes must not read them, or the scale would flush the live rows' values (or poison
    them).  (The depth cap of the weight-gradient form is exercised by the projection tests, whose pad rows
    leave stale rows behind the kept ones.)"""
    from gtg_vae_amd import functional as Fn

    torch.manual_seed(23)
    M, N, K, rows = 4224, 640, 384, 4100
    rd = torch.tensor([rows], dtype=torch.int32, device="cuda")
    A = torch.randn(M, K, device="cuda")
    B = torch.randn(N, K, device="cuda")
    A[rows:] = float("nan")
    A[rows:, ::2] = 3e37
    got = Fn.gemm_rows(A, B, True, True, rows_dev=rd, matrix_mode="f16x2")
    want = A[:rows].double() @ B.double().t()
    scale = A[:rows].double().abs() @ B.double().abs().t()
    assert ((got[:rows].double() - want).abs() / scale).max().item() < 2e-6
    assert bool((got[rows:] == 0).all())


def _unblock(p):
    """(3, K/16, R, 16) k-blocked planes -> (3, R, K)."""
    return p.permute(0, 2, 1, 3).reshape(3, p.shape[2], -1)


def test_split_planes_sum_is_exact_and_transposes():
    from gtg_vae_amd import functional as Fn

    torch.manual_seed(5)
    x = torch.randn(77, 128, device="cuda") * torch.logspace(-6, 6, 128, device="cuda")
    x[0, :2] = torch.tensor([0.0, -0.0], device="cuda")
    p = Fn.split_planes(x)
    assert p.shape == (3, 8, 77, 16) and p.dtype == torch.bfloat16
    u = _unblock(p).float()
    assert torch.equal(u[0] + u[1] + u[2], x)
    assert float((u[1].abs() > u[0].abs() * 2.0 ** -7).sum()) == 0       # each plane carries the next 8 bits
    xt = x.t().contiguous()                                             # (128, 77): the operand is its transpose
    pt = Fn.split_planes(xt, transpose=True)
    assert torch.equal(pt, p)
    view = torch.randn(40, 100, device="cuda")[:, 3:67]                 # row pitch larger than the width
    assert torch.equal(_unblock(Fn.split_planes(view)).float().sum(0), view)


@pytest.mark.parametrize("M,N,K,rows", [(128, 128, 16, None), (256, 384, 96, None), (300, 200, 48, None),
                                        (1000, 1024, 512, 777), (20759, 1024, 512, None), (4224, 512, 1024, 4100)])
@pytest.mark.parametrize("epi", [0, 3])
def test_gemm_on_presplit_planes_matches_float64_and_the_in_loop_split(M, N, K, rows, epi):
    """The GEMM on operands that arrive as three bf16 planes (global_load_lds staging, no split arithmetic in
    the loop): fp32-level error against float64, bit-identical to the GTG_MATRIX_BF16X3 GEMM that splits the
    same fp32 operands in its loop (same planes, same products, same order), ragged row counts read on the
    device and the tail split included."""
    from gtg_vae_amd import functional as Fn

    torch.manual_seed(M + N + K)
    A = torch.randn(M, K, device="cuda")
    W = torch.randn(N, K, device="cuda")
    aux = torch.randn(M, N, device="cuda") if epi == 3 else None
    rows_dev = None if rows is None else torch.tensor([rows], dtype=torch.int32, device="cuda")
    got = Fn.gemm_planes_rows(Fn.split_planes(A), Fn.split_planes(W), rows_dev, epilogue=epi, aux=aux)
    live = M if rows is None else rows
    want = A[:live].double() @ W.double().t()
    if epi == 3:
        want = torch.where(aux[:live] > 0, want, torch.zeros_like(want))
    err = (got[:live].double() - want).abs().max().item()
    assert err <= _tol(K) * max(want.abs().max().item(), 1.0) * 4, err
    assert float(got[live:].abs().max()) == 0.0 if live < M else True     # rows past the device count are not written
    if ((M + 127) // 128) * ((N + 127) // 128) >= 192:    # shapes the in-loop split also runs on 128x128 tiles
        same = Fn.gemm_rows(A, W, True, True, rows_dev, epilogue=epi, aux=aux, matrix_mode="bf16x3")
        assert torch.equal(got, same)
    # the transposed-weight image serves X * W with W stored (K, N)
    Wkn = W.t().contiguous()
    got2 = Fn.gemm_planes_rows(Fn.split_planes(A), Fn.split_planes(Wkn, transpose=True), rows_dev, epilogue=epi, aux=aux)
    assert torch.equal(got2, got)
