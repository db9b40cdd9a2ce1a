"""pxr_gemm_f32 against fp64 matmul for every tile variant (incl. the intra-workgroup split-K ones), operand flavour,
ragged edge and the split-K path -- the training step only exercises the heuristic's choices."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TILES = [0, 64, 128, 12864, 64128, 642, 3264]


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
