"""Thin torch-tensor wrappers over the C-ABI (include/pxr.h).  No arithmetic happens in Python here: each
function checks device/dtype/layout, allocates outputs through torch's caching allocator, and launches the HIP
kernel on torch's current stream.  Every function raises if the inputs are not CUDA(HIP) tensors -- there is
no CPU path.
"""
from __future__ import annotations

import os

import torch

from . import lib as _l

_CHECK_IDX = os.environ.get("PXR_CHECK_INDICES", "0") == "1"


def _req(t: torch.Tensor, dtype, name: str, contiguous: bool = True):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise _l.PxrError(f"{name}: expected a CUDA/HIP tensor (pixelrec_amd has no CPU fallback)")
    if t.dtype != dtype:
        raise _l.PxrError(f"{name}: expected dtype {dtype}, got {t.dtype}")
    if contiguous and not t.is_contiguous():
        raise _l.PxrError(f"{name}: expected a contiguous tensor")
    return t


class Workspace:
    """Grow-only scratch buffer (bytes) per device, reused by kernels that need temporary storage."""

    def __init__(self):
        self._buf = {}

    def get(self, nbytes: int, device) -> torch.Tensor:
        key = (device.type, device.index)
        b = self._buf.get(key)
        if b is None or b.numel() < nbytes:
            b = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self._buf[key] = b
        return b


_ws = Workspace()


# ------------------------------------------------------------------------------------------------ K1 gather
def embed_gather(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[..., :] = table[idx[...], :]  (reference: nn.Embedding forward, sasrec.py:68,101)."""
    L = _l.load()
    _req(table, torch.float32, "table")
    _req(idx, torch.int64, "idx")
    N, D = table.shape
    if _CHECK_IDX and idx.numel():
        lo, hi = int(idx.min()), int(idx.max())
        if lo < 0 or hi >= N:
            raise IndexError(f"index out of range in embed_gather: [{lo}, {hi}] vs N={N}")
    out = torch.empty(*idx.shape, D, dtype=torch.float32, device=table.device)
    _l.check(L.pxr_embed_gather_f32(_l.ptr(table), N, D, _l.ptr(idx), idx.numel(), _l.ptr(out), _l.stream_ptr()),
             "pxr_embed_gather_f32")
    return out


# ------------------------------------------------------------------------------------------------ GEMMs
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_MUL_DGELU = 0, 1, 2, 3


def gemm(a_kc: bool, b_kc: bool, M: int, N: int, K: int, A, lda, B, ldb, C, ldc, epilogue=EPI_NONE, bias=None,
         aux=None, ldaux=0, use_ws=True, tile_hint=0, split_hint=0):
    """Raw pxr_gemm_f32 (see gemm_f32.cuh for the operand flavours).  Used by tests and the bench."""
    L = _l.load()
    ws = None
    ws_bytes = 0
    if use_ws:
        ws_bytes = int(L.pxr_gemm_ws_bytes(int(a_kc), int(b_kc), M, N, K))
        ws_bytes = min(ws_bytes, 1 << 30)
        ws = _ws.get(ws_bytes, C.device)
    _l.check(L.pxr_gemm_f32(int(a_kc), int(b_kc), M, N, K, _l.ptr(A), lda, _l.ptr(B), ldb, _l.ptr(C), ldc, epilogue,
                            _l.ptr(bias), _l.ptr(aux), ldaux, _l.ptr(ws), ws_bytes, tile_hint, split_hint,
                            _l.stream_ptr()), "pxr_gemm_f32")
    return C


def linear_fwd(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor | None, gelu: bool = False):
    """y = x W^T + b (nn.Linear, layers.py:586-588,613,666,669).  gelu=True also returns the pre-activation."""
    L = _l.load()
    _req(x, torch.float32, "x"); _req(W, torch.float32, "W")
    N, K = W.shape
    M = x.numel() // K
    y = torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=x.device)
    pre = torch.empty_like(y) if gelu else None
    _l.check(L.pxr_linear_fwd_f32(_l.ptr(x), _l.ptr(W), _l.ptr(b), _l.ptr(y), _l.ptr(pre), M, N, K, int(gelu),
                                  _l.stream_ptr()), "pxr_linear_fwd_f32")
    return (y, pre) if gelu else y


def linear_bwd_input(dy: torch.Tensor, W: torch.Tensor, dgelu_pre: torch.Tensor | None = None):
    """dx = dy W  (optionally times gelu'(pre) of the layer that produced this linear's input)."""
    L = _l.load()
    _req(dy, torch.float32, "dy"); _req(W, torch.float32, "W")
    N, K = W.shape
    M = dy.numel() // N
    dx = torch.empty(*dy.shape[:-1], K, dtype=torch.float32, device=dy.device)
    _l.check(L.pxr_linear_bwd_input_f32(_l.ptr(dy), _l.ptr(W), _l.ptr(dx), _l.ptr(dgelu_pre), M, N, K,
                                        _l.stream_ptr()), "pxr_linear_bwd_input_f32")
    return dx


def linear_bwd_weight(dy: torch.Tensor, x: torch.Tensor, out: torch.Tensor | None = None):
    """dW = dy^T x   ([N,K], reduction over all tokens)."""
    L = _l.load()
    _req(dy, torch.float32, "dy"); _req(x, torch.float32, "x")
    N, K = dy.shape[-1], x.shape[-1]
    M = dy.numel() // N
    dW = out if out is not None else torch.empty(N, K, dtype=torch.float32, device=dy.device)
    ws_bytes = min(int(L.pxr_gemm_ws_bytes(0, 0, N, K, M)), 1 << 30)
    ws = _ws.get(ws_bytes, dy.device)
    _l.check(L.pxr_linear_bwd_weight_f32(_l.ptr(dy), _l.ptr(x), _l.ptr(dW), M, N, K, _l.ptr(ws), ws_bytes,
                                         _l.stream_ptr()), "pxr_linear_bwd_weight_f32")
    return dW


def colsum(x2d: torch.Tensor, out: torch.Tensor | None = None):
    """out[n] = sum_m x[m, n] (deterministic two-stage reduction)."""
    L = _l.load()
    _req(x2d, torch.float32, "x")
    M, N = x2d.shape
    o = out if out is not None else torch.empty(N, dtype=torch.float32, device=x2d.device)
    ws_bytes = int(L.pxr_colsum_ws_bytes(M, N))
    ws = _ws.get(ws_bytes, x2d.device)
    _l.check(L.pxr_colsum_f32(_l.ptr(x2d), N, M, N, _l.ptr(o), _l.ptr(ws), ws_bytes, _l.stream_ptr()), "pxr_colsum_f32")
    return o
