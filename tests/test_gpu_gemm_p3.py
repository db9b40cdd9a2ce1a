"""The planes GEMMs (pre-split bf16x3 operands in panel layout, csrc/gemm_p3.cuh) against fp64 and against the in-loop-split
kernels they replace (bit-identical products): forward (KC,KC), input gradient (KC,XC), grouped weight gradient (XC,XC) with
the bias gradient from the all-ones product, ragged shapes, output planes, the producers that write planes themselves
(LayerNorm sites, attention, flat AdamW) against a split of their fp32 output, and the non-finite operand behaviour."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TILES = [0, 406406430, 406406431, 412806420, 406412820, 812812830, 825612820, 425612833, 425612832, 425625631]
# the ping-pong tiles of gemm_p4.cuh: ...833 keeps gemm_p3's three accumulator sets (bit-identical), ...832 folds the small terms into
# one set and ...631 (256x256) runs ONE accumulator set: same products, different rounding points -- held to the fp64 bound only
NOT_BIT_IDENTICAL = {425612832, 425625631, 425612822}     # (...822: the weight gradient with the token reduction split in two)


def _tol(ref, K):
    return 2e-6 * (K ** 0.5) * float(ref.abs().max()) + 1e-6


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("M,N,K", [(3200, 512, 512), (333, 200, 96), (64, 64, 32), (1, 8, 32), (1025, 1028, 160), (130, 7, 64)])
def test_forward_matches_fp64_and_the_in_loop_split(tile, M, N, K):
    from pixelrec_amd import ops

    g = torch.Generator().manual_seed(M + N + K)
    x, W, b = torch.randn(M, K, generator=g), torch.randn(N, K, generator=g), torch.randn(N, generator=g)
    xs, Ws, bs = x.cuda(), W.cuda(), b.cuda()
    xp, Wp = ops.split_planes(xs), ops.split_planes(Ws)
    assert torch.equal(xp.to_dense(), xs) and torch.equal(Wp.to_dense(), Ws)            # the split is exact
    y = torch.full((M, N), float("nan"), device="cuda")
    ops.gemm_planes(xp, Wp, y, ops.EPI_BIAS, bias=bs, tile_hint=tile)
    ref = x.double() @ W.double().t() + b.double()
    assert (y.double().cpu() - ref).abs().max().item() <= _tol(ref, K)
    if N % 4 == 0:
        y0 = torch.empty(M, N, device="cuda")
        ops.gemm(True, True, M, N, K, xs, K, Ws, K, y0, N, ops.EPI_BIAS, bias=bs, use_ws=False, tile_hint=9064)
        if tile in NOT_BIT_IDENTICAL:
            assert (y - y0).abs().max().item() <= 0.05 * _tol(ref, K)
        else:
            assert torch.equal(y, y0)              # the planes ARE the terms gemm_b3 computes on the fly: same bits
    if N % 32 == 0:
        yp = ops.Planes.alloc(M, N, "cuda")
        ops.gemm_planes(xp, Wp, None, ops.EPI_BIAS, bias=bs, tile_hint=tile, Cp=yp)      # planes only, no fp32 store
        assert torch.equal(yp.to_dense(), y)


@pytest.mark.parametrize("tile", TILES)
@pytest.mark.parametrize("M,N,K", [(3200, 512, 1536), (333, 96, 224), (70, 32, 32), (1000, 1024, 512)])
def test_input_gradient_flavour(tile, M, N, K):
    """dX[M,N] = dY[M,K] W[K,N] (+ add | * mul): the B operand is read with transposing LDS loads."""
    from pixelrec_amd import ops

    g = torch.Generator().manual_seed(M * 3 + N)
    dy, W, add = torch.randn(M, K, generator=g).cuda(), torch.randn(K, N, generator=g).cuda(), torch.randn(M, N, generator=g).cuda()
    dyp, Wp = ops.split_planes(dy), ops.split_planes(W)
    for epi, aux in ((ops.EPI_NONE, None), (ops.EPI_ADD, add), (ops.EPI_MUL, add)):
        dx = torch.full((M, N), float("nan"), device="cuda")
        ops.gemm_planes(dyp, Wp, dx, epi, aux=aux, tile_hint=tile, b_kc=False)
        ref = dy.double().cpu() @ W.double().cpu()
        ref = ref + add.double().cpu() if epi == ops.EPI_ADD else (ref * add.double().cpu() if epi == ops.EPI_MUL else ref)
        assert (dx.double().cpu() - ref).abs().max().item() <= _tol(ref, K) * (4 if epi == ops.EPI_MUL else 1)
        dx0 = torch.empty(M, N, device="cuda")
        ops.gemm(True, False, M, N, K, dy, K, W, N, dx0, N, epi, aux=aux, ldaux=N, use_ws=False, tile_hint=9064)
        if tile in NOT_BIT_IDENTICAL:
            assert (dx - dx0).abs().max().item() <= 0.05 * _tol(ref, K) * (4 if epi == ops.EPI_MUL else 1)
        else:
            assert torch.equal(dx, dx0)


@pytest.mark.parametrize("tile", [0, 412812831, 812812830, 406406431, 425612832, 425612822])
def test_grouped_weight_gradient_and_bias(tile):
    from pixelrec_amd import ops

    g = torch.Generator().manual_seed(11)
    probs, pl, refs = [], [], []
    for T, N, K in ((3200, 512, 1024), (3200, 1536, 512), (80, 64, 96), (333, 32, 160)):
        dy, x = torch.randn(T, N, generator=g) * 0.01, torch.randn(T, K, generator=g)
        probs.append((dy.cuda(), x.cuda(), torch.empty(N, K, device="cuda"), torch.empty(N, device="cuda")))
        pl.append((ops.split_planes(dy.cuda()), ops.split_planes(x.cuda()), torch.full((N, K), float("nan"), device="cuda"),
                   torch.full((N,), float("nan"), device="cuda")))
        refs.append((dy.double().t() @ x.double(), dy.double().sum(0), T))
    ops.grouped_dw_planes(pl, tile_hint=tile)
    prev = ops.set_gemm_mode("bf16x3")
    try:
        ops.grouped_linear_bwd_weight(probs)
    finally:
        ops.set_gemm_mode(prev)
    for (_, _, dW, db), (_, _, dW0, db0), (rW, rb, T) in zip(pl, probs, refs):
        assert (dW.double().cpu() - rW).abs().max().item() <= _tol(rW, T)
        assert (db.double().cpu() - rb).abs().max().item() <= 2e-6 * (T ** 0.5) * float(rb.abs().max()) + 1e-7
        if tile in NOT_BIT_IDENTICAL:              # (two accumulator sets: same products, different rounding points)
            assert (dW - dW0).abs().max().item() <= 0.05 * _tol(rW, T)
        else:
            assert torch.equal(dW, dW0)            # same products, same order as grouped_dw_b3_kernel
    if tile == 425612822:                          # the split launch leaves its flags clean and its bits do not depend on the race
        again = [(a, b, torch.full_like(c, float("nan")), torch.full_like(d, float("nan"))) for a, b, c, d in pl]
        for _ in range(3):
            ops.grouped_dw_planes(again, tile_hint=tile)
            for (_, _, dW, db), (_, _, dW1, db1) in zip(pl, again):
                assert torch.equal(dW, dW1) and torch.equal(db, db1)
        ops.raise_on_bad_indices("cuda")


def test_producers_write_the_planes_of_their_fp32_output():
    """LayerNorm sites, LayerNorm backward, attention forward / backward and the flat AdamW write planes themselves: each
    must equal the exact split of the fp32 tensor the plain entry point writes (or of the updated weights)."""
    from pixelrec_amd import ops

    torch.manual_seed(3)
    B, L, D, H = 6, 20, 128, 4
    dev = "cuda"
    table, pos = torch.randn(300, D, device=dev), torch.randn(L, D, device=dev)
    gamma, beta = torch.rand(D, device=dev) + 0.5, torch.randn(D, device=dev)
    idx = torch.randint(0, 300, (B, L), device=dev)
    y, xh, rs, yp = ops.input_ln_fwd(table, idx, L, B, L, pos, gamma, beta, 1e-12, 0.1, 5, 0, planes=True)
    y0, _, _ = ops.input_ln_fwd(table, idx, L, B, L, pos, gamma, beta, 1e-12, 0.1, 5, 0)
    assert torch.equal(y, y0) and torch.equal(yp.to_dense(), y.view(B * L, D))
    x, res = torch.randn(B, L, D, device=dev), torch.randn(B, L, D, device=dev)
    y, xh, rs, yp = ops.ln_residual_fwd(x, res, gamma, beta, 1e-12, 0.2, 7, 3, planes=True)
    assert torch.equal(yp.to_dense(), y.view(B * L, D))
    dy = torch.randn(B, L, D, device=dev)
    dg, dbt = torch.empty(D, device=dev), torch.empty(D, device=dev)
    for p_drop in (0.0, 0.2):
        dz, dx, gp = ops.ln_bwd(0, dy, xh, rs, gamma, dg, dbt, p_drop, 7, 3, need_dx=p_drop > 0, planes=True)
        assert torch.equal(gp.to_dense(), (dx if dx is not None else dz).view(B * L, D))
    d = D // H
    qkv = torch.randn(B, L, 3 * D, device=dev)
    km = (torch.rand(B, L, device=dev) > 0.2).long()
    assert ops.attn_planes_supported(L, d)
    ctx, probs = ops.attn_fwd(qkv, km, L, B, H, L, d, 0.1, 9, 1)
    ctxp, probs2 = ops.attn_fwd(qkv, km, L, B, H, L, d, 0.1, 9, 1, planes=True)
    assert torch.equal(ctxp.to_dense(), ctx.view(B * L, D)) and torch.equal(probs, probs2)
    dctx = torch.randn(B, L, D, device=dev)
    dqkv = ops.attn_bwd(dctx, qkv, probs, B, H, L, d, 0.1, 9, 1)
    dqkvp = ops.attn_bwd(dctx, qkv, probs, B, H, L, d, 0.1, 9, 1, planes=True)
    assert torch.equal(dqkvp.to_dense(), dqkv.view(B * L, 3 * D))
    # split of several matrices in one launch
    mats = [torch.randn(96, 64, device=dev), torch.randn(33, 128, device=dev)]
    for m, p in zip(mats, ops.split_planes_multi(mats)):
        assert torch.equal(p.to_dense(), m)


def test_non_finite_operands_documented_behaviour():
    """csrc/gemm_b3.cuh / planes.cuh: hi = bf16(x) of an infinite (or > 3.39e38) operand is inf and the remainder x - hi is
    inf - inf = NaN, so an infinite operand poisons its products with NaN where torch.matmul (and the f32-input MFMA kernels)
    propagate +-inf; NaN operands give NaN in every mode.  Non-finite activations are an error state on this path either way
    (the trainer raises on a NaN loss like the reference, trainer.py:192-194): this test pins the difference down."""
    from pixelrec_amd import ops

    M, N, K = 64, 64, 64
    x, W = torch.ones(M, K, device="cuda"), torch.ones(N, K, device="cuda")
    x[3, 5] = float("inf")
    x[7, 0] = float("nan")
    y32, yb3, yp3 = (torch.empty(M, N, device="cuda") for _ in range(3))
    ops.gemm(True, True, M, N, K, x, K, W, K, y32, N, ops.EPI_NONE, use_ws=False, tile_hint=64)
    ops.gemm(True, True, M, N, K, x, K, W, K, yb3, N, ops.EPI_NONE, use_ws=False, tile_hint=9064)
    ops.gemm_planes(ops.split_planes(x), ops.split_planes(W), yp3, ops.EPI_NONE)
    ref = x @ W.t()
    assert torch.isinf(ref[3]).all() and torch.isinf(y32[3]).all()        # torch and the f32-input MFMA: +inf
    assert torch.isnan(yb3[3]).all() and torch.isnan(yp3[3]).all()        # bf16x3 (both forms): NaN
    for y in (y32, yb3, yp3):
        assert torch.isnan(y[7]).all()
        keep = [r for r in range(M) if r not in (3, 7)]
        assert torch.equal(y[keep], torch.full((M - 2, N), float(K), device="cuda"))     # other rows untouched
