"""pxr_gemm_f32 against fp64 matmul for every tile variant (incl. the intra-workgroup split-K ones), operand flavour,
ragged edge and the split-K path -- the training step only exercises the heuristic's choices."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TILES = [0, 64, 128, 12864, 64128, 642, 3264, 1281, 12861, 1282]


def _check(got, ref, K):
    tol = 2e-6 * (K ** 0.5) * float(ref.abs().max()) + 1e-6
    assert (got.double().cpu() - ref).abs().max().item() <= tol


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("M,N,K", [(3200, 512, 512), (77, 100, 36), (129, 193, 260), (64, 64, 32), (1, 4, 4)])
def test_forward_flavour_kc_kc(tile, M, N, K):
    from pixelrec_amd import ops

    g = torch.Generator().manual_seed(M * 7 + N)
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    y = torch.empty(M, N, device="cuda")
    ops.gemm(True, True, M, N, K, x.cuda(), K, W.cuda(), K, y, N, ops.EPI_BIAS, bias=b.cuda(), use_ws=False, tile_hint=tile)
    _check(y, x.double() @ W.double().t() + b.double(), K)


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("M,N,K", [(3200, 512, 1024), (76, 100, 52), (132, 196, 260)])
def test_input_grad_flavour_kc_xc(tile, M, N, K):
    from pixelrec_amd import ops

    g = torch.Generator().manual_seed(M + N * 3)
    dy, W, add = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g), torch.randn(M, N, generator=g)
    dx = torch.empty(M, N, device="cuda")
    ops.gemm(True, False, M, N, K, dy.cuda(), K, W.cuda(), N, dx, N, ops.EPI_ADD, aux=add.cuda(), ldaux=N, use_ws=False,
             tile_hint=tile)
    _check(dx, dy.double() @ W.double() + add.double(), K)


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("split", [0, 1, 3])
@pytest.mark.parametrize("M,N,K", [(512, 1024, 3200), (100, 76, 333 * 4), (68, 132, 40)])
def test_weight_grad_flavour_xc_xc(tile, split, M, N, K):
    from pixelrec_amd import ops

    g = torch.Generator().manual_seed(K + M)
    A, Bm = torch.randn(K, M, generator=g), torch.randn(K, N, generator=g)
    out = torch.empty(M, N, device="cuda")
    ops.gemm(False, False, M, N, K, A.cuda(), M, Bm.cuda(), N, out, N, ops.EPI_NONE, use_ws=True, tile_hint=tile,
             split_hint=split)
    _check(out, A.double().t() @ Bm.double(), K)


def test_gelu_epilogues_save_activation_or_its_derivative():
    """epilogue 2 saves the pre-activation (backward: 3 = *gelu'(aux)); epilogue 5 saves gelu'(pre-activation)
    (backward: 6 = *aux).  Both pairs give the same forward output and the same input gradient."""
    from pixelrec_amd import ops

    g = torch.Generator().manual_seed(5)
    M, N, K = 300, 256, 128
    x, W, b = torch.randn(M, K, generator=g).cuda(), torch.randn(N, K, generator=g).cuda() * 0.1, torch.randn(N, generator=g).cuda()
    y1, pre = ops.linear_fwd(x, W, b, gelu=True)
    y2, dg = ops.linear_fwd(x, W, b, gelu=True, save_grad=True)
    assert torch.equal(y1, y2)
    z = (x.double() @ W.double().t() + b.double()).cpu()
    ref_y = z * 0.5 * (1 + torch.erf(z / 2 ** 0.5))
    ref_dg = 0.5 * (1 + torch.erf(z / 2 ** 0.5)) + z * torch.exp(-0.5 * z * z) / (2 * torch.pi) ** 0.5
    assert (y1.cpu().double() - ref_y).abs().max().item() < 1e-5
    assert (pre.cpu().double() - z).abs().max().item() < 1e-5
    assert (dg.cpu().double() - ref_dg).abs().max().item() < 1e-5
    dy, W2 = torch.randn(M, 64, generator=g).cuda(), torch.randn(64, N, generator=g).cuda() * 0.1
    d1 = ops.linear_bwd_input(dy, W2, dgelu_pre=pre)
    d2 = ops.linear_bwd_input(dy, W2, mul=dg)
    assert torch.equal(d1, d2)
    assert (d1.cpu().double() - (dy.double().cpu() @ W2.double().cpu()) * ref_dg).abs().max().item() < 1e-5


@pytest.mark.parametrize("workers", [0, 16, 301, 1000])
@pytest.mark.parametrize("M,N,K", [(3200, 512, 512), (3200, 512, 2048), (3200, 2048, 512), (777, 132, 1024), (129, 196, 260),
                                   (64, 64, 128), (5, 8, 4096)])
def test_stream_k_forward_and_input_grad(workers, M, N, K):
    """The stream-K kernel (tile_hint 6464; split_hint = workers) for both operand flavours the step uses it with: right
    against fp64, and bit-identical from launch to launch (fixed summation order; the flags clean themselves)."""
    from pixelrec_amd import ops

    g = torch.Generator().manual_seed(M * 3 + N + K)
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    xs, Ws, bs = x.cuda(), W.cuda(), b.cuda()
    outs = []
    for _ in range(3):
        y = torch.full((M, N), float("nan"), device="cuda")
        pre = torch.empty(M, N, device="cuda")
        ops.gemm(True, True, M, N, K, xs, K, Ws, K, y, N, ops.EPI_BIAS_GELU, bias=bs, aux=pre, ldaux=N, use_ws=False,
                 tile_hint=6464, split_hint=workers)
        outs.append((y, pre))
    ref = x.double() @ W.double().t() + b.double()
    _check(outs[0][1], ref, K)
    _check(outs[0][0], torch.nn.functional.gelu(ref), K)
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][0], outs[2][0]) and torch.equal(outs[0][1], outs[2][1])
    # dX = dY W + residual-branch gradient  (KC, XC)
    Wt, add = torch.randn(K, N, generator=g), torch.randn(M, N, generator=g)
    dx = [torch.full((M, N), float("nan"), device="cuda") for _ in range(2)]
    for d in dx:
        ops.gemm(True, False, M, N, K, xs, K, Wt.cuda(), N, d, N, ops.EPI_ADD, aux=add.cuda(), ldaux=N, use_ws=False,
                 tile_hint=6464, split_hint=workers)
    _check(dx[0], x.double() @ Wt.double() + add.double(), K)
    assert torch.equal(dx[0], dx[1])
    assert int(ops.device_status("cuda").item()) & 8 == 0


def test_stream_k_on_two_streams_at_once():
    """Each stream has its own partial-tile scratch: launches on two streams may overlap."""
    from pixelrec_amd import ops

    g = torch.Generator().manual_seed(3)
    M, N, K = 3200, 512, 1024
    x, W = torch.randn(M, K, generator=g).cuda(), torch.randn(N, K, generator=g).cuda()
    ref = torch.empty(M, N, device="cuda")
    ops.gemm(True, True, M, N, K, x, K, W, K, ref, N, ops.EPI_NONE, use_ws=False, tile_hint=6464)
    torch.cuda.synchronize()
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    outs = [[torch.empty(M, N, device="cuda") for _ in range(8)] for _ in range(2)]
    for i in range(8):
        for s, o in ((s1, outs[0][i]), (s2, outs[1][i])):
            with torch.cuda.stream(s):
                ops.gemm(True, True, M, N, K, x, K, W, K, o, N, ops.EPI_NONE, use_ws=False, tile_hint=6464)
    torch.cuda.synchronize()
    for o in outs[0] + outs[1]:
        assert torch.equal(o, ref)
