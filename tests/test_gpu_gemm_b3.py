"""The fp32 GEMM on the bf16 matrix pipe (exact 3 x bf16 operand split, csrc/gemm_b3.cuh) against fp64, held to the SAME
tolerance as the fp32-MFMA kernel (tests/test_gpu_gemm.py::_check) and compared with that kernel's own error."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _err(got, ref):
    return (got.double().cpu() - ref).abs().max().item()


def _tol(ref, K):
    return 2e-6 * (K ** 0.5) * float(ref.abs().max()) + 1e-6


def test_split_is_exact():
    """hi + mid + lo == x bit for bit (normal range), both orientations, padded columns are zero."""
    from pixelrec_amd import ops

    g = torch.Generator().manual_seed(0)
    x = (torch.randn(300, 52, generator=g) * torch.exp(4 * torch.randn(300, 52, generator=g))).cuda()
    x[0, :4] = torch.tensor([0.0, 1.0, -3.0e-20, 65504.0])
    for tr in (False, True):
        pl = ops.Planes(*(x.t().shape if tr else x.shape), "cuda")
        ops.split_planes([(x, tr, pl)])
        p = pl.buf.view(3, pl.R, pl.ldd).float()
        want = x.t() if tr else x
        assert torch.equal(p[0, :, :pl.C] + p[1, :, :pl.C] + p[2, :, :pl.C], want)      # fp32 adds of the three terms are exact here
        assert torch.equal((p[0, :, :pl.C].double() + p[1, :, :pl.C].double() + p[2, :, :pl.C].double()).float(), want)
        assert float(p[:, :, pl.C:].abs().sum()) == 0.0
        assert (p[1].abs() <= p[0].abs() * 2.0 ** -8 + 1e-30).all() and (p[2].abs() <= p[0].abs() * 2.0 ** -16 + 1e-30).all()


@pytest.mark.parametrize("tile", [0, 64, 641, 642, 643, 1281, 12811])
@pytest.mark.parametrize("M,N,K", [(3200, 512, 512), (3200, 1536, 512), (3200, 512, 1536), (77, 100, 36), (129, 196, 260),
                                   (64, 64, 32), (1, 4, 8), (1024, 4099, 512)])
def test_forward_and_input_grad_against_fp64(tile, M, N, K):
    from pixelrec_amd import ops

    g = torch.Generator().manual_seed(M * 7 + N + K)
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    xs, Ws, bs = x.cuda(), W.cuda(), b.cuda()
    pl = ops.Planes(N, K, "cuda")
    ops.split_planes([(Ws, False, pl)])
    y = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_b3(M, N, K, xs, K, pl, y, N, ops.EPI_BIAS, bias=bs, tile_hint=tile)
    ref = x.double() @ W.double().t() + b.double()
    e_b3 = _err(y, ref)
    y32 = torch.empty(M, N, device="cuda")
    ops.gemm(True, True, M, N, K, xs, K, Ws, K, y32, N, ops.EPI_BIAS, bias=bs, use_ws=False)
    e_f32 = _err(y32, ref)
    assert e_b3 <= _tol(ref, K), (e_b3, e_f32)
    assert e_b3 <= 4.0 * e_f32 + 1e-6, (e_b3, e_f32)          # the same error class as the exact-fp32 MFMA chain
    if N % 4:
        return
    # dX = dY W + add through planes of W^T
    dy, add = torch.randn(M, N, generator=g).cuda(), torch.randn(M, K, generator=g).cuda()
    plt = ops.Planes(K, N, "cuda")
    ops.split_planes([(Ws, True, plt)])
    dx = torch.full((M, K), float("nan"), device="cuda")
    ops.gemm_b3(M, K, N, dy, N, plt, dx, K, ops.EPI_ADD, aux=add, ldaux=K, tile_hint=tile)
    refx = dy.double().cpu() @ W.double() + add.double().cpu()
    assert _err(dx, refx) <= _tol(refx, N)
    y2 = torch.empty_like(y)
    ops.gemm_b3(M, N, K, xs, K, pl, y2, N, ops.EPI_BIAS, bias=bs, tile_hint=tile)
    assert torch.equal(y, y2)                                 # deterministic


def test_wide_dynamic_range_and_epilogues():
    """Operands spanning 12 orders of magnitude (gradients next to activations), GELU epilogue pair."""
    from pixelrec_amd import ops

    g = torch.Generator().manual_seed(9)
    M, N, K = 333, 200, 96
    x = torch.randn(M, K, generator=g) * torch.exp(7 * torch.randn(M, 1, generator=g))
    W = torch.randn(N, K, generator=g) * torch.exp(7 * torch.randn(N, 1, generator=g)) * 1e-3
    ref = x.double() @ W.double().t()
    pl = ops.Planes(N, K, "cuda")
    ops.split_planes([(W.cuda(), False, pl)])
    y = torch.empty(M, N, device="cuda")
    ops.gemm_b3(M, N, K, x.cuda(), K, pl, y, N)
    scale = x.double().abs() @ W.double().abs().t()             # per-element error scale sum |a||b|
    assert ((y.double().cpu() - ref).abs() / scale).max().item() < 2.0 ** -21
    xs = torch.randn(M, K, generator=g).cuda()
    Ws, b = (torch.randn(N, K, generator=g) * 0.1).cuda(), torch.randn(N, generator=g).cuda()
    ops.split_planes([(Ws, False, pl)])
    yg, dg = torch.empty(M, N, device="cuda"), torch.empty(M, N, device="cuda")
    ops.gemm_b3(M, N, K, xs, K, pl, yg, N, ops.EPI_BIAS_GELU_GRAD, bias=b, aux=dg, ldaux=N)
    v = xs.double() @ Ws.double().t() + b.double()
    assert (yg.double() - torch.nn.functional.gelu(v)).abs().max().item() < 1e-5
    cdf = 0.5 * (1 + torch.erf(v / 2 ** 0.5))
    assert (dg.double() - (cdf + v * torch.exp(-0.5 * v * v) / (2 * torch.pi) ** 0.5)).abs().max().item() < 1e-5
