"""Torch (CPU) stand-ins for the pixelrec_amd.ops entry points that model/vit_native.py calls -- TEST INFRASTRUCTURE.

They let the CPU suite check the ORCHESTRATION of the native image encoder (operand offsets / strides of the batched
attention GEMMs, the backward formulas, the flat parameter packing) against torch autograd without a GPU.  The kernels
themselves are checked on the GPU (tests/test_gpu_mosasrec.py, tests/test_gpu_configs.py, tests/test_gpu_vit.py)."""
import torch

EPI_BIAS_ADD, EPI_BIAS_QGELU_GRAD, EPI_BIAS_RELU = 7, 8, 9


def ln_residual_fwd(x, res, gamma, beta, eps, p_drop=0.0, seed=0, stream_id=0, save=True, step_dev=None):
    z = x if res is None else x + res
    mean = z.mean(-1, keepdim=True)
    var = ((z - mean) ** 2).mean(-1, keepdim=True)
    rstd = 1.0 / torch.sqrt(var + eps)
    xhat = (z - mean) * rstd
    return xhat * gamma + beta, (xhat if save else None), (rstd.reshape(-1) if save else None)


def ln_bwd(gather_mode, dy, xhat, rstd, gamma, dgamma, dbeta, p_drop=0.0, seed=0, stream_id=0, need_dx=False,
           step_dev=None, defer=None):
    D = dy.shape[-1]
    g = dy * gamma
    dz = (g - g.mean(-1, keepdim=True) - xhat * (g * xhat).mean(-1, keepdim=True)) * rstd.view(*dy.shape[:-1], 1)
    dgamma.copy_((dy * xhat).reshape(-1, D).sum(0))
    dbeta.copy_(dy.reshape(-1, D).sum(0))
    return dz, None


def linear_fwd(x, W, b, gelu=False, save_grad=False):
    y = x @ W.t()
    return y + b if b is not None else y


def linear_epi(x, W, b, epi, aux=None, tag=""):
    v = x @ W.t() + b
    if epi == EPI_BIAS_ADD:
        return v + aux
    if epi == EPI_BIAS_QGELU_GRAD:
        s = torch.sigmoid(1.702 * v)
        return v * s, s + 1.702 * v * s * (1 - s)
    if epi == EPI_BIAS_RELU:
        return torch.relu(v)
    raise ValueError(epi)


def linear_bwd_input(dy, W, dgelu_pre=None, add=None, mul=None):
    dx = dy @ W
    if mul is not None:
        dx = dx * mul
    if add is not None:
        dx = dx + add
    return dx


def linear_bwd_weight(dy, x, out=None):
    out.copy_(dy.reshape(-1, dy.shape[-1]).t() @ x.reshape(-1, x.shape[-1]))
    return out


def grouped_linear_bwd_weight(problems):
    for dy, x, dW, db in problems:
        dW.copy_(dy.t() @ x)
        if db is not None:
            db.copy_(dy.sum(0))


def _operand(t, off, ld, rows, cols, z, nb2, s12):
    base = off + (z // nb2) * s12[0] + (z % nb2) * s12[1]
    return torch.as_strided(t.reshape(-1), (rows, cols), (ld, 1), base)


def gemm_batched(a_kc, b_kc, M, N, K, A, a_off, lda, B, b_off, ldb, C, c_off, ldc, batch, nb2, a12, b12, c12):
    for z in range(batch):
        a = _operand(A, a_off, lda, M, K, z, nb2, a12) if a_kc else _operand(A, a_off, lda, K, M, z, nb2, a12).t()
        b = _operand(B, b_off, ldb, N, K, z, nb2, b12).t() if b_kc else _operand(B, b_off, ldb, K, N, z, nb2, b12)
        _operand(C, c_off, ldc, M, N, z, nb2, c12).copy_(a @ b)


def softmax_rows(S, rows, T, ld, scale):
    v = S.view(rows, ld)
    p = torch.softmax(v[:, :T] * scale, dim=-1)
    v.zero_()
    v[:, :T] = p


def softmax_rows_bwd(P, dP, rows, T, ld, scale):
    p, d = P.view(rows, ld)[:, :T], dP.view(rows, ld)
    ds = scale * p * (d[:, :T] - (d[:, :T] * p).sum(-1, keepdim=True))
    d.zero_()
    d[:, :T] = ds


def vit_embed(patches, cls, pos):
    n = patches.shape[0]
    return torch.cat([cls.expand(n, 1, -1), patches], dim=1) + pos[None]


def token_mean(x):
    return x.mean(dim=1)


def token_mean_relu_bwd(dout, act):
    return (act > 0).float() * dout[:, None, :] / act.shape[1]


def add(a, b):
    return a + b


def colsum(x2d, out=None, defer=None):
    out.copy_(x2d.sum(0))
    return out


class DeferredReductions:
    def flush(self, bump=None):
        return False


def install(monkeypatch):
    """Point pixelrec_amd.ops (as seen by vit_native) at the stand-ins and let the packing accept CPU tensors."""
    from pixelrec_amd import ops
    from pixelrec_amd.model import vit_native

    for name in ("ln_residual_fwd", "ln_bwd", "linear_fwd", "linear_epi", "linear_bwd_input", "linear_bwd_weight",
                 "grouped_linear_bwd_weight", "gemm_batched", "softmax_rows", "softmax_rows_bwd", "vit_embed", "token_mean",
                 "token_mean_relu_bwd", "add", "colsum", "DeferredReductions"):
        monkeypatch.setattr(ops, name, globals()[name])

    # the stand-ins cover the fp32-operand orchestration; the planes / fused-attention variants of the same blocks only exist
    # as HIP kernels and are checked on the GPU (tests/test_gpu_vit.py, test_gpu_configs.py, test_gpu_tower_attn.py)
    monkeypatch.setenv("PXR_PLANES", "0")
    monkeypatch.setenv("PXR_TOWER_ATTN", "0")
    monkeypatch.setattr(vit_native.NativeTower, "_require_hip", staticmethod(lambda dev: None))


# ---- row-sharded table helpers (csrc/embed_grad.hip: shard_* / scatter_rows kernels), restated in torch for the CPU test of
# the exchange choreography (tests/test_sharded_exchange_gloo.py).  Same contracts as the pixelrec_amd.ops wrappers.
def shard_bucket_ids(ids, n_dev, world, n_table, pp_cap, pad_id):
    n = int(n_dev[0])
    req = torch.full((world, pp_cap), pad_id, dtype=torch.int64)
    pos = torch.full((world, pp_cap), -1, dtype=torch.int32)
    counts = torch.zeros(world, dtype=torch.int32)
    for i in range(n):
        v = int(ids[i])
        if 0 < v < n_table:
            o = v % world
            j = int(counts[o])
            if j < pp_cap:
                req[o, j] = v
                pos[o, j] = i
            counts[o] += 1
    return req, pos, counts


def shard_local_rows(ids, world, rank, n_table):
    ok = (ids > 0) & (ids < n_table) & (ids % world == rank)
    return torch.where(ok, ids // world + 1, torch.zeros_like(ids))


def shard_first_rows(ids_all, world, rank, n_table):
    cap = ids_all.numel() // world
    out = shard_local_rows(ids_all, world, rank, n_table).clone()
    seen = set()
    for q in range(world):
        for j in range(cap):
            t = q * cap + j
            if int(out[t]):
                v = int(ids_all[t])
                if v in seen:
                    out[t] = 0
                seen.add(v)
    return out


def embed_gather(table, idx):
    return table[idx]


def scatter_rows(src, pos, dst, row_offset=0):
    keep = pos >= 0
    dst[pos[keep].long() + row_offset] = src[keep]
    return dst
