"""Fused tower attention (csrc/tower_attn.hip) against an fp64 restatement of HF CLIPAttention.forward (what the item tower of
REC/model/modules.py runs per block: softmax(q k^T / sqrt(d)) v, no mask, no dropout)."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def _ref(qkv, n, T, heads, d, dtype=torch.float64):
    H = heads * d
    x = qkv.to(dtype).view(n, T, 3, heads, d)
    k, v, q = x[:, :, 0], x[:, :, 1], x[:, :, 2]                  # the packed projection is k | v | q
    s = torch.einsum("bthc,bshc->bhts", q, k) * d ** -0.5
    p = torch.softmax(s, dim=-1)
    ctx = torch.einsum("bhts,bshc->bthc", p, v).reshape(n, T, H)
    return ctx, torch.logsumexp(s, dim=-1).reshape(n * heads, T)


@pytest.mark.parametrize("T", [1, 5, 32, 33, 50, 96, 197, 257, 288])
@pytest.mark.parametrize("spread", [1.0, 6.0])
def test_tower_attn_fwd_matches_fp64(T, spread):
    from pixelrec_amd import ops
    n, heads, d = 3, 5, 64
    assert ops.tower_attn_supported(T, d)
    g = torch.Generator().manual_seed(100 + T)
    qkv = (torch.randn(n * T, 3 * heads * d, generator=g) * spread).cuda()
    ctx, planes, lse = ops.tower_attn_fwd(qkv, n, T, heads, d, 2 * heads * d, 0, heads * d, d ** -0.5, ctx=True, planes=True, lse=True)
    ref, ref_lse = _ref(qkv.cpu(), n, T, heads, d)
    scale = float(ref.abs().max())
    # fp32-grade: every product is carried to ~2^-24 and accumulated in fp32.  With peaky softmaxes (spread 6: scaled scores
    # of +-150) the fp32 rounding of the scores themselves dominates, so the yardstick is what plain fp32 arithmetic achieves
    ref32 = _ref(qkv.cpu(), n, T, heads, d, torch.float32)[0]
    err32 = float((ref32.double() - ref).abs().max())
    assert float((ctx.cpu().double() - ref).abs().max()) < max(3e-6 * scale, 3.0 * err32)
    assert float((lse.cpu().double() - ref_lse).abs().max()) < 2e-5 * max(1.0, float(ref_lse.abs().max()))
    # the planes are the exact split of the same fp32 values
    assert torch.equal(planes.to_dense(), ctx.view(n * T, heads * d))


def test_tower_attn_rejects_unsupported_shapes():
    from pixelrec_amd import ops
    assert not ops.tower_attn_supported(289, 64)
    assert not ops.tower_attn_supported(197, 32)
    qkv = torch.zeros(2 * 10, 3 * 32, device="cuda")
    with pytest.raises(RuntimeError):
        ops.tower_attn_fwd(qkv, 2, 10, 1, 32, 64, 0, 32, 1.0)


@pytest.mark.parametrize("T", [1, 5, 33, 64, 197, 257])
@pytest.mark.parametrize("spread", [1.0, 4.0])
def test_tower_attn_bwd_matches_fp64_autograd(T, spread):
    """pxr_tower_attn_bwd_f32 (probabilities recomputed from the forward's log-sum-exp, nothing T x T in memory) against
    torch autograd of the fp64 restatement; yardstick = what fp32 autograd achieves on the same inputs."""
    from pixelrec_amd import ops
    n, heads, d = 2, 3, 64
    H = heads * d
    g = torch.Generator().manual_seed(500 + T)
    qkv = torch.randn(n * T, 3 * H, generator=g) * spread
    dctx = torch.randn(n, T, H, generator=g)

    def grads(dtype):
        x = qkv.to(dtype).requires_grad_(True)
        ctx, _ = _ref(x, n, T, heads, d, dtype)
        (ctx * dctx.to(dtype)).sum().backward()
        return x.grad

    ref, ref32 = grads(torch.float64), grads(torch.float32)
    q = qkv.cuda()
    ctx, _, lse = ops.tower_attn_fwd(q, n, T, heads, d, 2 * H, 0, H, d ** -0.5, ctx=True, planes=False, lse=True)
    got = ops.tower_attn_bwd(q, dctx.cuda().view(n * T, H), ctx.view(n * T, H), lse, n, T, heads, d, 2 * H, 0, H, d ** -0.5)
    assert got.shape == qkv.shape and bool(torch.isfinite(got).all())
    scale = float(ref.abs().max())
    err32 = float((ref32.double() - ref).abs().max())
    err = float((got.cpu().double() - ref).abs().max())
    assert err < max(3e-6 * scale, 3.0 * err32), (err, err32, scale)
    for name, lo in (("k", 0), ("v", H), ("q", 2 * H)):           # every one of the three ranges carries signal
        if T > 1 or name == "v":                                  # (one token: the softmax is constant, dq = dk = 0 exactly)
            assert float(ref[:, lo:lo + H].abs().max()) > 0 and float(got[:, lo:lo + H].abs().max()) > 0, name
        else:
            assert float(got[:, lo:lo + H].abs().max()) < 1e-5 * scale, name      # rounding of 1 - exp(s - lse) and dP - delta
