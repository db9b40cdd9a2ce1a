"""The fp32 GEMM on the bf16 matrix pipe (exact 3 x bf16 operand split, csrc/gemm_b3.cuh) against fp64: every operand
flavour, both tiles, ragged edges, split-K, batched -- held to the SAME tolerance as the f32-input MFMA kernel
(tests/test_gpu_gemm.py::_check) and compared with that kernel's own error."""
import pytest
import torch

pytestmark = pytest.mark.gpu

B3_TILES = [9064, 91281]


def _err(got, ref):
    return (got.double().cpu() - ref).abs().max().item()


def _tol(ref, K):
    return 2e-6 * (K ** 0.5) * float(ref.abs().max()) + 1e-6


@pytest.mark.parametrize("tile", B3_TILES)
@pytest.mark.parametrize("M,N,K", [(3200, 512, 512), (3200, 1536, 512), (3200, 512, 1536), (77, 100, 36), (129, 196, 260),
                                   (64, 64, 32), (1, 4, 4), (1024, 4100, 512)])
def test_forward_and_input_grad_against_fp64(tile, M, N, K):
    from pixelrec_amd import ops

    g = torch.Generator().manual_seed(M * 7 + N + K)
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    xs, Ws, bs = x.cuda(), W.cuda(), b.cuda()
    y = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm(True, True, M, N, K, xs, K, Ws, K, y, N, ops.EPI_BIAS, bias=bs, use_ws=False, tile_hint=tile)
    ref = x.double() @ W.double().t() + b.double()
    e_b3 = _err(y, ref)
    y32 = torch.empty(M, N, device="cuda")
    ops.gemm(True, True, M, N, K, xs, K, Ws, K, y32, N, ops.EPI_BIAS, bias=bs, use_ws=False, tile_hint=64)
    e_f32 = _err(y32, ref)
    assert e_b3 <= _tol(ref, K), (e_b3, e_f32)
    assert e_b3 <= 4.0 * e_f32 + 1e-6, (e_b3, e_f32)          # the same error class as the exact-fp32 MFMA chain
    # dX = dY W + add:  (KC, XC)
    dy, add = torch.randn(M, N, generator=g).cuda(), torch.randn(M, K, generator=g).cuda()
    dx = torch.full((M, K), float("nan"), device="cuda")
    ops.gemm(True, False, M, K, N, dy, N, Ws, K, dx, K, ops.EPI_ADD, aux=add, ldaux=K, use_ws=False, tile_hint=tile)
    refx = dy.double().cpu() @ W.double() + add.double().cpu()
    assert _err(dx, refx) <= _tol(refx, N)
    y2 = torch.empty_like(y)
    ops.gemm(True, True, M, N, K, xs, K, Ws, K, y2, N, ops.EPI_BIAS, bias=bs, use_ws=False, tile_hint=tile)
    assert torch.equal(y, y2)                                 # deterministic


@pytest.mark.parametrize("tile", B3_TILES)
@pytest.mark.parametrize("split", [0, 1, 3])
@pytest.mark.parametrize("M,N,K", [(512, 1024, 3200), (100, 76, 333 * 4), (68, 132, 40), (33, 7, 50)])
def test_weight_grad_flavour_xc_xc(tile, split, M, N, K):
    from pixelrec_amd import ops

    if split and N % 4:
        pytest.skip("split-K needs N % 4 == 0")
    g = torch.Generator().manual_seed(K + M)
    A, Bm = torch.randn(K, M + (-M) % 4, generator=g), torch.randn(K, N + (-N) % 4, generator=g)
    out = torch.empty(M, N, device="cuda")
    ops.gemm(False, False, M, N, K, A.cuda(), A.shape[1], Bm.cuda(), Bm.shape[1], out, N, ops.EPI_NONE, use_ws=bool(split != 1),
             tile_hint=tile, split_hint=split)
    ref = A[:, :M].double().t() @ Bm[:, :N].double()
    assert _err(out, ref) <= _tol(ref, K)


def test_wide_dynamic_range_and_epilogues():
    """Operands spanning 12 orders of magnitude (gradients next to activations), GELU epilogue pair, default mode."""
    from pixelrec_amd import ops

    prev = ops.set_gemm_mode("bf16x3")
    try:
        _wide_range_body(ops)
    finally:
        ops.set_gemm_mode(prev)


def _wide_range_body(ops):
    g = torch.Generator().manual_seed(9)
    M, N, K = 333, 200, 96
    x = torch.randn(M, K, generator=g) * torch.exp(7 * torch.randn(M, 1, generator=g))
    W = torch.randn(N, K, generator=g) * torch.exp(7 * torch.randn(N, 1, generator=g)) * 1e-3
    ref = x.double() @ W.double().t()
    y = torch.empty(M, N, device="cuda")
    ops.gemm(True, True, M, N, K, x.cuda(), K, W.cuda(), K, y, N, use_ws=False)
    scale = x.double().abs() @ W.double().abs().t()             # per-element error scale sum |a||b|
    assert ((y.double().cpu() - ref).abs() / scale).max().item() < 2.0 ** -21
    xs = torch.randn(M, K, generator=g).cuda()
    Ws, b = (torch.randn(N, K, generator=g) * 0.1).cuda(), torch.randn(N, generator=g).cuda()
    yg, dg = ops.linear_fwd(xs, Ws, b, gelu=True, save_grad=True)
    v = xs.double() @ Ws.double().t() + b.double()
    assert (yg.double() - torch.nn.functional.gelu(v)).abs().max().item() < 1e-5
    cdf = 0.5 * (1 + torch.erf(v / 2 ** 0.5))
    assert (dg.double() - (cdf + v * torch.exp(-0.5 * v * v) / (2 * torch.pi) ** 0.5)).abs().max().item() < 1e-5


def test_grouped_weight_grads_and_mode_switch():
    """All dW + db of a backward pass in one launch, bf16x3 against f32 mode and fp64; ragged shapes."""
    from pixelrec_amd import ops

    g = torch.Generator().manual_seed(4)
    shapes = [(3200, 1536, 512), (3200, 512, 512), (3200, 1024, 512), (3200, 512, 1024), (130, 68, 36)]
    probs = []
    for M, N, K in shapes:
        probs.append((torch.randn(M, N, generator=g).cuda(), torch.randn(M, K, generator=g).cuda()))
    outs = {}
    for mode in ("bf16x3", "f32"):
        prev = ops.set_gemm_mode(mode)
        try:
            res = [(torch.empty(dy.shape[1], x.shape[1], device="cuda"), torch.empty(dy.shape[1], device="cuda")) for dy, x in probs]
            ops.grouped_linear_bwd_weight([(dy, x, dW, db) for (dy, x), (dW, db) in zip(probs, res)])
            outs[mode] = res
        finally:
            ops.set_gemm_mode(prev)
    for (dy, x), (dW, db), (dW32, db32) in zip(probs, outs["bf16x3"], outs["f32"]):
        ref = dy.double().t() @ x.double()
        assert _err(dW, ref.cpu()) <= _tol(ref.cpu(), dy.shape[0])
        assert _err(dW, ref.cpu()) <= 4.0 * _err(dW32, ref.cpu()) + 1e-6
        refb = dy.double().sum(0).cpu()
        assert _err(db, refb) <= 1e-5 * max(1.0, float(refb.abs().max()))
