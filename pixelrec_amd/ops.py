"""Thin torch-tensor wrappers over the C-ABI (include/pxr.h).  No arithmetic happens in Python here: each
function checks device/dtype/layout, allocates outputs through torch's caching allocator, and launches the HIP
kernel on torch's current stream.  Every function raises if the inputs are not CUDA(HIP) tensors -- there is
no CPU path.
"""
from __future__ import annotations

import ctypes
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
    """Grow-only scratch buffer (bytes) per (device, stream), reused by kernels that need temporary storage.
    Keyed by the CURRENT stream so that work issued on a side stream never shares scratch with the main stream."""

    def __init__(self):
        self._buf = {}
        self._retired = []

    def get(self, nbytes: int, device) -> torch.Tensor:
        device = torch.device(device)
        stream = _l.raw_stream() if device.type == "cuda" else 0
        key = (device.type, device.index, stream)
        b = self._buf.get(key)
        if b is None or b.numel() < nbytes:
            if b is not None:
                self._retired.append(b)   # a captured hipGraph may still hold this pointer: never free it
            b = torch.empty(max(int(nbytes), 1 << 20), dtype=torch.uint8, device=device)
            self._buf[key] = b
        return b


_ws = Workspace()

# bench.py: set to a list to collect (start_event, end_event, work, tag) around kernel launches: every fp32-MFMA GEMM
# launch (work = flops) and the tagged HBM-side kernels (lazy AdamW rows, fused gather+LN, dense table sweep; work =
# algorithmic bytes, 0 when it is data dependent).  Events are recorded on the stream the kernel is launched on.
GEMM_TIMING = None


class _gemm_timer:
    def __init__(self, flops, tag="gemm"):
        self.flops = flops
        self.tag = tag          # names the kernel instantiation the launch resolves to (see bench.py's roofline)
        self.on = GEMM_TIMING is not None
        if self.on and tag.startswith(("gemm_kernel", "grouped_dw_kernel")) and _l.load().pxr_get_gemm_mode():
            # bf16x3 GEMM mode: the same launch sites resolve to the bf16-pipe kernels (csrc/gemm_b3.hip)
            self.tag = tag.replace("gemm_kernel", "gemm_b3_kernel", 1).replace("grouped_dw_kernel", "grouped_dw_b3_kernel", 1)

    def __enter__(self):
        if self.on:
            self.ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            self.ev[0].record()

    def __exit__(self, *exc):
        if self.on:
            self.ev[1].record()
            GEMM_TIMING.append((self.ev[0], self.ev[1], self.flops, self.tag))


# ------------------------------------------------------------------------------------------------ bad-index flag
_status = {}


def device_status(device=None) -> torch.Tensor:
    """The int32 device status word of this process' GPU (created and registered with the library on first use)."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    if dev.index is None:            # "cuda" without an index names the current device, not a second status word
        dev = torch.device("cuda", torch.cuda.current_device())
    t = _status.get(dev.index)
    if t is None:
        t = _status[dev.index] = torch.zeros(1, dtype=torch.int32, device=dev)
        _l.check(_l.load().pxr_set_status_word(_l.ptr(t)), "pxr_set_status_word")
    return t


STATUS_SHARD_OVERFLOW = 16      # PXR_STATUS_SHARD_OVERFLOW (csrc/pxr_common.h)


def clear_status_bits(device, bits: int):
    """Drop `bits` from the device status word (stream-ordered; for a condition the caller has just handled itself)."""
    if torch.device(device).type != "cuda":
        return
    device_status(device).bitwise_and_(~int(bits))


class StatusPoll:
    """The status word watched WITHOUT stalling the stream: start() enqueues a 4-byte copy into pinned host memory behind the work
    issued so far; check() looks at the copy of an EARLIER start() once its event has completed (a query, no wait) and raises through
    raise_on_bad_indices when a bit is set.  A condition is therefore reported at most two polling intervals after the kernel that
    flagged it (graph.GraphedTrainStep polls every PXR_H2_REFRESH_STEPS replays), not at the end of the epoch."""

    def __init__(self, device=None):
        self.device = device
        self._host = torch.zeros(1, dtype=torch.int32).pin_memory()
        self._ev = None

    def start(self):
        self._host.copy_(device_status(self.device), non_blocking=True)
        self._ev = torch.cuda.Event()
        self._ev.record()

    def check(self):
        if self._ev is None or not self._ev.query():
            return
        self._ev = None
        if int(self._host[0]):
            raise_on_bad_indices(self.device)


def raise_on_bad_indices(device=None):
    """Host check of the status word (synchronises): IndexError if a gather kernel met an item id outside the table
    since the last check -- what nn.Embedding raises in the reference (sasrec.py:68).  Called where the host
    synchronises anyway: once per training epoch (with the loss), after predict / evaluate, by embed_gather under
    PXR_CHECK_INDICES=1."""
    t = device_status(device)
    v = int(t.item())
    if v:
        t.zero_()
    if v & 8:
        _l.load().pxr_gemm_reset_flags()     # the flag words are in an unknown state: zero them before anything launches again
        raise RuntimeError("stream-K / split-K GEMM: a worker timed out waiting for a partial tile (results are wrong; the flag "
                           "words were reset); set PXR_GEMM_SK=0 PXR_DW_SPLITK=0 and report")
    if v & 1:
        raise IndexError("index out of range in self (an item id outside [0, item_num) reached an embedding gather)")
    if v & 4:
        raise RuntimeError("fused scoring + top-k: the per-user candidate buffer of the threshold pass overflowed (results may miss "
                           "items); with PXR_TOPK_PRODUCTS=1 first go back to 3 or 6 (a tighter threshold margin admits fewer "
                           "candidates), else set PXR_TOPK_VARIANT=2 (register lists, no candidate buffer)")
    if v & 32:
        raise RuntimeError("fused scoring + top-k: a user ended with fewer than K candidates above its threshold (non-finite scores "
                           "or embeddings?); the affected rows of the result hold id -1")
    if v & 64:
        raise RuntimeError("fp16 two-plane operands (image tower, or the sequence block of batches >= PXR_SEQ_H2_MIN_TOKENS "
                           "tokens): an activation, a gradient or a scaled weight was NaN or left the fp16 range (|x| > 65504) "
                           "-- the affected outputs are inf / nan; set PXR_TOWER_H2=0 and / or PXR_SEQ_H2=0 (the six-product "
                           "bf16x3 GEMMs have fp32's range)")
    if v & 128:
        raise H2StaleOverflow("fp16 two-plane gradients under the recent steps' scale (PXR_SEQ_H2_STALE): a gradient exceeded the "
                              f"decaying maximum of the last steps more than 2^{H2_STALE_HEADROOM + 2}-fold; its largest elements were "
                              "saturated in that step (no inf / nan was written).  The Trainer goes on with per-step exact scales "
                              "(PXR_SEQ_H2_STALE=0: six more launches per step)")
    if v & 16:
        raise RuntimeError("row-sharded table: one rank owned more of a batch's hit rows than the per-pair request capacity "
                           "(ShardedSASRec.pair_slack); rows were dropped -- raise the slack or use row_exchange='reduce_scatter'")
    if v & 2:
        raise RuntimeError("data-parallel row exchange: a rank's batch touched more unique table rows than the configured "
                           "exchange capacity (GradSync(exchange_rows=...)); gradient rows were dropped -- raise the bound")


# ------------------------------------------------------------------------------------------------ K1 gather
def embed_gather(table: torch.Tensor, idx: torch.Tensor) -> torch.Tensor:
    """out[..., :] = table[idx[...], :]  (reference: nn.Embedding forward, sasrec.py:68,101)."""
    L = _l.load()
    _req(table, torch.float32, "table")
    _req(idx, torch.int64, "idx")
    N, D = table.shape
    device_status(table.device)       # out-of-range ids are flagged on the device (and clamped), raised at the next check
    out = torch.empty(*idx.shape, D, dtype=torch.float32, device=table.device)
    _l.check(L.pxr_embed_gather_f32(_l.ptr(table), N, D, _l.ptr(idx), idx.numel(), _l.ptr(out), _l.stream_ptr()),
             "pxr_embed_gather_f32")
    if _CHECK_IDX:
        raise_on_bad_indices(table.device)
    return out


# ------------------------------------------------------------------------------------------------ GEMMs
EPI_NONE, EPI_BIAS, EPI_BIAS_GELU, EPI_MUL_DGELU, EPI_ADD = 0, 1, 2, 3, 4
EPI_BIAS_GELU_GRAD, EPI_MUL, EPI_BIAS_ADD, EPI_BIAS_QGELU_GRAD, EPI_BIAS_RELU, EPI_BIAS_ACT_GRAD = 5, 6, 7, 8, 9, 10
EPI_BIAS_QGELU = 11         # planes GEMMs only: quick_gelu(x W^T + b), no derivative


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
    with _gemm_timer(2.0 * M * N * K):
        _l.check(L.pxr_gemm_f32(int(a_kc), int(b_kc), M, N, K, _l.ptr(A), lda, _l.ptr(B), ldb, _l.ptr(C), ldc,
                                epilogue, _l.ptr(bias), _l.ptr(aux), ldaux, _l.ptr(ws), ws_bytes, tile_hint,
                                split_hint, _l.stream_ptr()), "pxr_gemm_f32")
    return C


class Planes:
    """An fp32 matrix [rows, cols] pre-split into planes in PANEL layout (include/pxr.h, csrc/planes.cuh): the operand format of
    gemm_planes.  fmt 0: three bf16 planes (x = hi + mid + lo exactly); fmt 1 ("h2"): two fp16 planes of x * 2^exp (22 significant
    bits; the forward-only blocks of the image tower).  `buf` is a flat 2-byte-element tensor; `off` / `pr` / `ps` are the element
    offset of plane 0, the rows per panel and the plane stride, so row / column ranges are views of the same buffer."""
    __slots__ = ("buf", "rows", "cols", "pr", "ps", "off", "fmt", "exp", "exp_dev", "stats", "pre_bound")

    def __init__(self, buf, rows, cols, pr, ps, off=0, fmt=0, exp=0, exp_dev=None, stats=None):
        self.buf, self.rows, self.cols, self.pr, self.ps, self.off, self.fmt, self.exp = buf, rows, cols, pr, ps, off, fmt, exp
        # fmt 1 with a scale chosen on the device (split_h2_auto / a bound): int32 [1] holding the exponent (then `exp` is unused)
        # and float32 [2] = (max |x|, for weights the largest column sum of |x|)
        self.exp_dev, self.stats = exp_dev, stats
        self.pre_bound = None      # (W.stats address, factor, int32 [1]): an h2_bound_exp its split launch already computed

    @staticmethod
    def alloc(rows: int, cols: int, device, fmt: int = 0) -> "Planes":
        if cols % 32:
            raise _l.PxrError(f"planes need a multiple of 32 columns, got {cols}")
        pr = (rows + 31) // 32 * 32
        ps = pr * cols
        # rows past `rows` of a panel are read by edge tiles and, when the rows are a GEMM's reduction dimension (weight
        # gradients), multiplied in: they must be zero.  Producers never write them, so one fill at allocation is enough.
        mk = torch.empty if pr == rows else torch.zeros
        if fmt:
            return Planes(mk(2 * ps, dtype=torch.float16, device=device), rows, cols, pr, ps, fmt=1)
        return Planes(mk(3 * ps, dtype=torch.bfloat16, device=device), rows, cols, pr, ps)

    def ptr(self):
        return ctypes.c_void_p(self.buf.data_ptr() + 2 * self.off)

    def row_range(self, r0: int, r1: int) -> "Planes":
        assert r0 % 16 == 0 and r0 <= r1 <= self.rows
        return Planes(self.buf, r1 - r0, self.cols, self.pr, self.ps, self.off + 32 * r0, self.fmt, self.exp, self.exp_dev, self.stats)

    def col_range(self, c0: int, c1: int) -> "Planes":
        assert c0 % 32 == 0 and c1 % 32 == 0 and c0 <= c1 <= self.cols
        return Planes(self.buf, self.rows, c1 - c0, self.pr, self.ps, self.off + (c0 // 32) * self.pr * 32, self.fmt, self.exp,
                      self.exp_dev, self.stats)

    def to_dense(self) -> torch.Tensor:
        """fp32 [rows, cols] = the sum of the planes (exact for fmt 0; x rounded to 22 bits for fmt 1).  For tests; not on the
        product path."""
        dev = self.buf.device
        r = torch.arange(self.rows, device=dev).view(-1, 1)
        c = torch.arange(self.cols, device=dev).view(1, -1)
        idx = ((c // 32) * self.pr + r) * 32 + ((((c // 8) % 4) ^ ((r // 4) % 4)) * 8) + c % 8 + self.off
        out = None
        for q in range(2 if self.fmt else 3):
            t = self.buf[(idx + q * self.ps).reshape(-1)].view(self.rows, self.cols).float()
            out = t if out is None else out + t
        if self.fmt:
            out = out * (2.0 ** -(int(self.exp_dev.item()) if self.exp_dev is not None else self.exp))
        return out


def split_planes(x: torch.Tensor, out: Planes | None = None) -> Planes:
    """fp32 [rows, cols] -> Planes with x = hi + mid + lo exactly (pxr_split_planes_f32)."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32
    rows, cols = x.shape
    if out is None:
        out = Planes.alloc(rows, cols, x.device)
    assert (out.rows, out.cols) == (rows, cols)
    _l.check(_l.load().pxr_split_planes_f32(_l.ptr(x), rows, cols, x.stride(0), out.ptr(), out.ps, out.pr,
                                            _l.stream_ptr()), "pxr_split_planes_f32")
    return out


def gemm_planes(Ap: Planes, Bp: Planes, C: torch.Tensor | None, epilogue=EPI_NONE, bias=None, aux=None, act=0, tile_hint=0,
                b_kc=True, Cp: Planes | None = None):
    """C[M,N] = A[M,K] x B from planes (pxr_gemm_planes_f32): Bp is [N,K] (b_kc) or [K,N]; C (fp32) and / or Cp (planes of
    the result) are written."""
    M, K = Ap.rows, Ap.cols
    N = Bp.rows if b_kc else Bp.cols
    assert (Bp.cols if b_kc else Bp.rows) == K
    assert C is None or C.shape == (M, N)
    if Ap.fmt or Bp.fmt:
        # two fp16 planes per operand, three products per multiply (pxr_gemm_h2_f32)
        if not (Ap.fmt and Bp.fmt):
            raise _l.PxrError("gemm_planes: fp16 two-plane operands need BOTH operands in that format")
        device_status(Ap.buf.device)      # the producers' fp16 range check reports through the status word
        with _gemm_timer(2.0 * M * N * K, "gemm_p3_kernel<P4Cfg<..., HALF>> (fp16 two-plane operands, 3 products)"):
            _l.check(_l.load().pxr_gemm_h2_f32(int(b_kc), M, N, K, Ap.ptr(), Ap.ps, Ap.pr, Ap.exp, _l.ptr(Ap.exp_dev), Bp.ptr(), Bp.ps,
                                               Bp.pr, Bp.exp, _l.ptr(Bp.exp_dev), _l.ptr(C), C.stride(0) if C is not None else 0,
                                               epilogue, _l.ptr(bias), _l.ptr(aux), aux.stride(0) if aux is not None else 0, *_pl(Cp),
                                               Cp.fmt if Cp is not None else 0,
                                               _l.ptr(Cp.exp_dev) if (Cp is not None and Cp.fmt) else None, act, tile_hint,
                                               _l.stream_ptr()), "pxr_gemm_h2_f32")
        return C
    assert Cp is None or Cp.fmt == 0, "the bf16x3 GEMMs write bf16x3 planes"
    with _gemm_timer(2.0 * M * N * K):
        _l.check(_l.load().pxr_gemm_planes_f32(int(b_kc), M, N, K, Ap.ptr(), Ap.ps, Ap.pr, Bp.ptr(), Bp.ps, Bp.pr, _l.ptr(C),
                                               C.stride(0) if C is not None else 0, epilogue, _l.ptr(bias), _l.ptr(aux),
                                               aux.stride(0) if aux is not None else 0,
                                               Cp.ptr() if Cp is not None else None, Cp.ps if Cp is not None else 0,
                                               Cp.pr if Cp is not None else 0, act, tile_hint, _l.stream_ptr()),
                 "pxr_gemm_planes_f32")
    return C


def set_gemm_mode(mode: str):
    """"bf16x3" (default; PXR_GEMM_MODE) = the heuristic's GEMM choices run on the bf16 matrix pipe through the exact
    3 x bf16 operand split (csrc/gemm_b3.cuh), "f32" = on the f32-input MFMA.  Process-wide; returns the previous mode."""
    L = _l.load()
    prev = "bf16x3" if L.pxr_get_gemm_mode() else "f32"
    if mode not in ("bf16x3", "f32"):
        raise ValueError(f"gemm mode must be 'bf16x3' or 'f32', got {mode!r}")
    _l.check(L.pxr_set_gemm_mode(1 if mode == "bf16x3" else 0), "pxr_set_gemm_mode")
    return prev


def gemm_mode() -> str:
    return "bf16x3" if _l.load().pxr_get_gemm_mode() else "f32"


ACT_CODES = {"relu": 3, "swish": 4, "tanh": 5, "sigmoid": 6}      # the reference's ACT2FN besides gelu (layers.py:642-649)


def linear_fwd(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor | None, gelu: bool = False, save_grad: bool = False,
               act: str | None = None):
    """y = x W^T + b (nn.Linear, layers.py:586-588,613,666,669).  gelu=True also returns the pre-activation, or with
    save_grad=True gelu'(pre-activation) (what linear_bwd_input(mul=...) multiplies by).  act in ACT_CODES: that
    activation instead of erf-GELU, always with its derivative returned second."""
    if act is not None and act != "gelu":
        gelu, code = True, ACT_CODES[act]
    else:
        code = None
    L = _l.load()
    _req(x, torch.float32, "x"); _req(W, torch.float32, "W")
    N, K = W.shape
    M = x.numel() // K
    y = torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=x.device)
    pre = torch.empty_like(y) if gelu else None
    with _gemm_timer(2.0 * M * N * K, "gemm_kernel<KC,KC,EPI_BIAS_GELU*> (fwd + erf-GELU)" if gelu else
                     "gemm_kernel<KC,KC,EPI_BIAS> (fwd)"):
        _l.check(L.pxr_linear_fwd_f32(_l.ptr(x), _l.ptr(W), _l.ptr(b), _l.ptr(y), _l.ptr(pre), M, N, K,
                                      code if code is not None else ((2 if save_grad else 1) if gelu else 0),
                                      _l.stream_ptr()), "pxr_linear_fwd_f32")
    return (y, pre) if gelu else y


def linear_epi(x: torch.Tensor, W: torch.Tensor, b: torch.Tensor, epi: int, aux: torch.Tensor | None = None,
               tag: str = "gemm_kernel<KC,KC,*> (ViT linear)"):
    """y = epilogue(x W^T + b) for the ViT-block epilogues (gemm_f32.hip): EPI_BIAS_ADD (aux = residual stream, read),
    EPI_BIAS_QGELU_GRAD (aux = quick_gelu' saved for the backward, written; returned second), EPI_BIAS_RELU."""
    L = _l.load()
    _req(x, torch.float32, "x"); _req(W, torch.float32, "W")
    N, K = W.shape
    M = x.numel() // K
    y = torch.empty(*x.shape[:-1], N, dtype=torch.float32, device=x.device)
    out_aux = None
    if epi == EPI_BIAS_QGELU_GRAD:
        aux = out_aux = torch.empty_like(y)
    elif epi == EPI_BIAS_ADD:
        _req(aux, torch.float32, "aux")
    with _gemm_timer(2.0 * M * N * K, tag):
        _l.check(L.pxr_gemm_f32(1, 1, M, N, K, _l.ptr(x), K, _l.ptr(W), K, _l.ptr(y), N, epi, _l.ptr(b), _l.ptr(aux), N,
                                None, 0, 0, 0, _l.stream_ptr()), "pxr_gemm_f32")
    return (y, out_aux) if out_aux is not None else y


def gemm_batched(a_kc: bool, b_kc: bool, M: int, N: int, K: int, A, a_off, lda, B, b_off, ldb, C, c_off, ldc, batch, nb2,
                 a12, b12, c12):
    """`batch` GEMMs in one launch (pxr_gemm_batched_f32).  A/B/C are tensors, *_off element offsets of problem 0,
    x12 = (stride1, stride2) in elements for the two batch levels (z // nb2, z % nb2)."""
    L = _l.load()
    cp = _l.c_void_p
    with _gemm_timer(2.0 * M * N * K * batch, "gemm_kernel (batched: attention contraction of the ViT encoder)"):
        _l.check(L.pxr_gemm_batched_f32(int(a_kc), int(b_kc), M, N, K, cp(A.data_ptr() + 4 * a_off), lda,
                                        cp(B.data_ptr() + 4 * b_off), ldb, cp(C.data_ptr() + 4 * c_off), ldc, batch, nb2,
                                        a12[0], a12[1], b12[0], b12[1], c12[0], c12[1], 0, _l.stream_ptr()),
                 "pxr_gemm_batched_f32")


def tower_attn_supported(T: int, d: int) -> bool:
    return os.environ.get("PXR_TOWER_ATTN", "1") != "0" and bool(_l.load().pxr_tower_attn_supported(int(T), int(d)))


def tower_attn_fwd(qkv: torch.Tensor, n: int, T: int, heads: int, d: int, q_off: int, k_off: int, v_off: int, scale: float,
                   ctx: bool = True, planes: bool = False, lse: bool = False):
    """Fused attention of a tower block (pxr_tower_attn_fwd_f32) on the fused projection output qkv [n*T, ld]: q / k / v start at
    columns q_off / k_off / v_off.  Returns (ctx fp32 [n, T, heads*d] | None, ctx Planes | None, lse [n*heads, T] | None)."""
    _req(qkv, torch.float32, "qkv")
    ld = qkv.shape[-1]
    H = heads * d
    out = torch.empty(n, T, H, dtype=torch.float32, device=qkv.device) if ctx else None
    h2 = planes == "h2"               # context planes in the two-plane fp16 format (the operand of the h2 out-projection GEMM)
    op = Planes.alloc(n * T, H, qkv.device, fmt=int(h2)) if planes else None
    ls = torch.empty(n * heads, T, dtype=torch.float32, device=qkv.device) if lse else None
    cp = _l.c_void_p
    base = qkv.data_ptr()
    fn = _l.load().pxr_tower_attn_fwd_h2_f32 if h2 else _l.load().pxr_tower_attn_fwd_f32
    if h2:
        device_status(qkv.device)
    with _gemm_timer(4.0 * T * T * d * n * heads, "tower_attn_fwd_kernel (fused QK^T / softmax / PV of a tower block)"):
        _l.check(fn(cp(base + 4 * q_off), cp(base + 4 * k_off), cp(base + 4 * v_off), ld, n, heads, T,
                                                  d, float(scale), _l.ptr(out) if ctx else None, H, *_pl(op),
                                                  _l.ptr(ls) if lse else None, _l.stream_ptr()), "pxr_tower_attn_fwd_f32")
    return out, op, ls


def tower_attn_bwd(qkv: torch.Tensor, dctx: torch.Tensor, ctx: torch.Tensor, lse: torch.Tensor, n: int, T: int, heads: int, d: int,
                   q_off: int, k_off: int, v_off: int, scale: float) -> torch.Tensor:
    """Backward of tower_attn_fwd (pxr_tower_attn_bwd_f32): -> dqkv with the layout of qkv (every column of the q / k / v
    ranges is written).  ctx / lse: the forward's fp32 context and log-sum-exp."""
    _req(qkv, torch.float32, "qkv"); _req(dctx, torch.float32, "dctx"); _req(ctx, torch.float32, "ctx"); _req(lse, torch.float32, "lse")
    ld, H = qkv.shape[-1], heads * d
    dqkv = torch.empty_like(qkv)
    ws = torch.empty(n * heads * T, dtype=torch.float32, device=qkv.device)
    cp = _l.c_void_p
    b, g = qkv.data_ptr(), dqkv.data_ptr()
    with _gemm_timer(10.0 * T * T * d * n * heads, "tower_attn_bwd_{dq,dkv}_kernel (fused attention backward of a tower block)"):
        _l.check(_l.load().pxr_tower_attn_bwd_f32(cp(b + 4 * q_off), cp(b + 4 * k_off), cp(b + 4 * v_off), ld, _l.ptr(dctx),
                                                  _l.ptr(ctx), H, _l.ptr(lse), n, heads, T, d, float(scale), cp(g + 4 * q_off),
                                                  cp(g + 4 * k_off), cp(g + 4 * v_off), ld, _l.ptr(ws), _l.stream_ptr()),
                 "pxr_tower_attn_bwd_f32")
    return dqkv


def causal_im2col(x: torch.Tensor, k: int, dilation: int) -> torch.Tensor:
    """x [B, L, C] -> xcol [B, L, C*k] with xcol[b, t, c*k + j] = x[b, t - (k-1-j)*dilation, c] (pxr_causal_im2col_f32)."""
    _req(x, torch.float32, "x")
    B, L, C = x.shape
    out = torch.empty(B, L, C * k, dtype=torch.float32, device=x.device)
    _l.check(_l.load().pxr_causal_im2col_f32(_l.ptr(x), _l.ptr(out), B, L, C, k, dilation, _l.stream_ptr()), "pxr_causal_im2col_f32")
    return out


def causal_col2im(dxcol: torch.Tensor, k: int, dilation: int) -> torch.Tensor:
    """The transpose of causal_im2col: dxcol [B, L, C*k] -> dx [B, L, C] (pxr_causal_col2im_f32)."""
    _req(dxcol, torch.float32, "dxcol")
    B, L, Ck = dxcol.shape
    out = torch.empty(B, L, Ck // k, dtype=torch.float32, device=dxcol.device)
    _l.check(_l.load().pxr_causal_col2im_f32(_l.ptr(dxcol), _l.ptr(out), B, L, Ck // k, k, dilation, _l.stream_ptr()),
             "pxr_causal_col2im_f32")
    return out


def gru_gates_fwd(gi, gh, h_prev, h_out, save=None):
    """One GRU step's gates (pxr_gru_gates_fwd_f32): gi, gh [B, 3H]; h_prev [B, H] | None; writes h_out [B, H] (and save [B, 4H])."""
    B, H = h_out.shape
    _l.check(_l.load().pxr_gru_gates_fwd_f32(_l.ptr(gi), _l.ptr(gh), _l.ptr(h_prev), _l.ptr(h_out), _l.ptr(save), B, H,
                                             _l.stream_ptr()), "pxr_gru_gates_fwd_f32")


def gru_gates_bwd(dh, save, h_prev, dgi, dgh, dh_prev):
    """Backward of gru_gates_fwd (pxr_gru_gates_bwd_f32): dh [B, H] -> dgi, dgh [B, 3H], dh_prev [B, H] = dh * z."""
    B, H = dh.shape
    _l.check(_l.load().pxr_gru_gates_bwd_f32(_l.ptr(dh), _l.ptr(save), _l.ptr(h_prev), _l.ptr(dgi), _l.ptr(dgh), _l.ptr(dh_prev),
                                             B, H, _l.stream_ptr()), "pxr_gru_gates_bwd_f32")


def softmax_rows(S: torch.Tensor, rows: int, T: int, ld: int, scale: float):
    _l.check(_l.load().pxr_softmax_rows_f32(_l.ptr(S), rows, T, ld, float(scale), _l.stream_ptr()), "pxr_softmax_rows_f32")


def softmax_rows_bwd(P: torch.Tensor, dP: torch.Tensor, rows: int, T: int, ld: int, scale: float):
    _l.check(_l.load().pxr_softmax_rows_bwd_f32(_l.ptr(P), _l.ptr(dP), rows, T, ld, float(scale), _l.stream_ptr()),
             "pxr_softmax_rows_bwd_f32")


def vit_embed(patches: torch.Tensor, cls: torch.Tensor, pos: torch.Tensor) -> torch.Tensor:
    """patches [n, T-1, H] -> tokens [n, T, H] = [cls | patches] + pos."""
    _req(patches, torch.float32, "patches"); _req(cls, torch.float32, "cls"); _req(pos, torch.float32, "pos")
    n, Tm1, H = patches.shape
    out = torch.empty(n, Tm1 + 1, H, dtype=torch.float32, device=patches.device)
    _l.check(_l.load().pxr_vit_embed_f32(_l.ptr(patches), _l.ptr(cls), _l.ptr(pos), _l.ptr(out), n, Tm1 + 1, H,
                                         _l.stream_ptr()), "pxr_vit_embed_f32")
    return out


def token_mean(x: torch.Tensor) -> torch.Tensor:
    _req(x, torch.float32, "x")
    n, T, D = x.shape
    out = torch.empty(n, D, dtype=torch.float32, device=x.device)
    _l.check(_l.load().pxr_token_mean_f32(_l.ptr(x), _l.ptr(out), n, T, D, _l.stream_ptr()), "pxr_token_mean_f32")
    return out


def token_mean_relu_bwd(dout: torch.Tensor, act: torch.Tensor) -> torch.Tensor:
    _req(dout, torch.float32, "dout"); _req(act, torch.float32, "act")
    n, T, D = act.shape
    dact = torch.empty_like(act)
    _l.check(_l.load().pxr_token_mean_relu_bwd_f32(_l.ptr(dout), _l.ptr(act), _l.ptr(dact), n, T, D, _l.stream_ptr()),
             "pxr_token_mean_relu_bwd_f32")
    return dact


def add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    _req(a, torch.float32, "a"); _req(b, torch.float32, "b")
    out = torch.empty_like(a)
    _l.check(_l.load().pxr_add_f32(_l.ptr(a), _l.ptr(b), _l.ptr(out), a.numel(), _l.stream_ptr()), "pxr_add_f32")
    return out


def dropout(x: torch.Tensor, p: float, seed: int, stream_id: int, step_dev=None) -> torch.Tensor:
    """x / (1 - p) where the library's counter hash keeps the element, else 0 (pxr_dropout_f32); on the upstream gradient with the
    same (seed, stream_id, step_dev) it is the backward of itself."""
    _req(x, torch.float32, "x")
    y = torch.empty_like(x)
    _l.check(_l.load().pxr_dropout_f32(_l.ptr(x), _l.ptr(y), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream_id),
                                       _l.ptr(step_dev), _l.stream_ptr()), "pxr_dropout_f32")
    return y


def linear_bwd_input(dy: torch.Tensor, W: torch.Tensor, dgelu_pre: torch.Tensor | None = None,
                     add: torch.Tensor | None = None, mul: torch.Tensor | None = None):
    """dx = dy W, optionally times gelu'(pre) (from the saved pre-activation), times `mul` (gelu' saved by the forward),
    or plus `add` (residual gradient)."""
    L = _l.load()
    _req(dy, torch.float32, "dy"); _req(W, torch.float32, "W")
    N, K = W.shape
    M = dy.numel() // N
    dx = torch.empty(*dy.shape[:-1], K, dtype=torch.float32, device=dy.device)
    tag = ("gemm_kernel<KC,XC,EPI_MUL_DGELU> (dX through GELU)" if dgelu_pre is not None else
           "gemm_kernel<KC,XC,EPI_MUL> (dX x saved gelu')" if mul is not None else
           "gemm_kernel<KC,XC,EPI_ADD> (dX + residual grad)" if add is not None else "gemm_kernel<KC,XC,EPI_NONE> (dX)")
    with _gemm_timer(2.0 * M * N * K, tag):
        _l.check(L.pxr_linear_bwd_input_f32(_l.ptr(dy), _l.ptr(W), _l.ptr(dx), _l.ptr(dgelu_pre), _l.ptr(add),
                                            _l.ptr(mul), M, N, K, _l.stream_ptr()), "pxr_linear_bwd_input_f32")
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
    with _gemm_timer(2.0 * M * N * K, "gemm_kernel<XC,XC> (dW, split-K)"):
        _l.check(L.pxr_linear_bwd_weight_f32(_l.ptr(dy), _l.ptr(x), _l.ptr(dW), M, N, K, _l.ptr(ws), ws_bytes,
                                             _l.stream_ptr()), "pxr_linear_bwd_weight_f32")
    return dW


def grouped_linear_bwd_weight(problems):
    """problems: list of (dy2d [M,N], x2d [M,K], dW [N,K] out, db [N] out | None).  One launch for all of them (one per MULTI_MAX
    of them: blocks deeper than 4 layers)."""
    import ctypes

    if len(problems) > MULTI_MAX:
        for lo in range(0, len(problems), MULTI_MAX):
            grouped_linear_bwd_weight(problems[lo:lo + MULTI_MAX])
        return
    if len(problems) == 1:
        # ONE small weight under a long token reduction (the image tower's head: a 512 x 768 weight over 69 344 tokens is 24 tiles
        # of 128 x 128, each 69 344 tokens deep -- 1.5 ms on a tenth of the chip): the token range is cut into parts that run as
        # the problems of ONE grouped launch into a workspace, and a column-sum launch adds the parts in part order
        dy, x, dW, db = problems[0]
        M, N, K = dy.shape[0], dy.shape[1], x.shape[1]
        S = _dw_token_parts(M, N, K)
        # (N % 4: part i starts i * (N K + N) floats into the workspace -- 16-byte aligned for the kernels' vector accesses only then)
        if S > 1 and N % 4 == 0 and K % 4 == 0 and dy.is_contiguous() and x.is_contiguous() and dW.is_contiguous():
            step = (-(-M // S) + 63) // 64 * 64
            S = -(-M // step)
            # one workspace per (stream, shape): the side stream of fork_pending / overlap_weight_grads may run this route beside
            # the main stream's
            row = N * K + N
            key = (dy.device, _l.raw_stream(), S, N, K)
            ws = _dw_split_ws.get(key)
            if ws is None:
                if len(_dw_split_ws) >= 8:       # a handful of (stream, shape) pairs exist in practice: never grow without bound
                    _dw_split_ws.clear()
                ws = _dw_split_ws[key] = torch.empty(S, row, dtype=torch.float32, device=dy.device)
            grouped_linear_bwd_weight([(dy[i * step:(i + 1) * step], x[i * step:(i + 1) * step], ws[i, :N * K].view(N, K), ws[i, N * K:])
                                       for i in range(S)])
            red = DeferredReductions()
            red.add(ws, S, N * K + N, dW, db if db is not None else ws.new_empty(N), N * K)
            red.flush()
            return
    L = _l.load()
    n = len(problems)
    P, I = ctypes.c_void_p * n, ctypes.c_int * n
    dy = P(*[p[0].data_ptr() for p in problems])
    x = P(*[p[1].data_ptr() for p in problems])
    dW = P(*[p[2].data_ptr() for p in problems])
    db = P(*[(p[3].data_ptr() if p[3] is not None else None) for p in problems])
    M = I(*[p[0].shape[0] for p in problems])
    N = I(*[p[0].shape[1] for p in problems])
    K = I(*[p[1].shape[1] for p in problems])
    flops = sum(2.0 * p[0].shape[0] * p[0].shape[1] * p[1].shape[1] for p in problems)
    with _gemm_timer(flops, "grouped_dw_kernel (all dW + db of the step)"):
        _l.check(L.pxr_grouped_linear_bwd_weight_f32(n, dy, x, dW, db, M, N, K, _l.stream_ptr()),
                 "pxr_grouped_linear_bwd_weight_f32")


_dw_split_ws: dict = {}


def _dw_token_parts(M: int, N: int, K: int) -> int:
    """Parts of the token range for a single weight gradient (grouped_linear_bwd_weight): 1 unless its 128 x 128 tiles cover less
    than a third of the chip under a reduction of >= 16 384 tokens; then enough parts for 192 tiles, each >= 2 048 tokens deep."""
    if os.environ.get("PXR_DW_TOKEN_SPLIT", "1") == "0":
        return 1
    t128 = ((N + 127) // 128) * ((K + 127) // 128)
    if M < 16384 or t128 >= 96:
        return 1
    return max(1, min(MULTI_MAX, -(-192 // t128), M // 2048))


MULTI_MAX = 16      # matrices per pxr_split_planes_multi_f32 launch / plane segments per pxr_adamw_flat_tab_planes_f32 launch


def h2_exponent(max_abs: float) -> int:
    """The power-of-two exponent e that puts max_abs * 2^e into [2^13, 2^14): a factor 4 below the fp16 maximum, 38 binades above
    its smallest subnormal (csrc/planes.cuh "h2")."""
    import math

    if not (max_abs > 0.0) or math.isinf(max_abs):
        return 0
    return max(-60, min(60, 14 - math.frexp(max_abs)[1]))


def split_h2_auto(mats, col_stats: bool = False, stats: torch.Tensor | None = None, outs=None, with_buffers: bool = False, top: int = 0):
    """fp32 matrices -> h2 Planes whose power-of-two scales are chosen ON THE DEVICE (pxr_h2_split_auto_multi_f32): no host
    synchronisation -- gradients, weights that an optimizer step just moved.  Each result carries `exp_dev` (int32 [1]) and
    `stats` (float32 [2]: max |x|; then the largest column sum of |x| with col_stats, else rows * max |x| -- what h2_bound_exp needs
    of a weight).  `stats` ([len(mats), 2] float32): the producers of the matrices already gathered max |x| into stats[:, 0]
    (ln_bwd / attn_bwd with stat=...): no statistics pass at all.  top (8..15; default 14): the largest scaled value lands in
    [2^(top-1), 2^top) -- more headroom below the fp16 limit for tensors rewritten in place with the same exponent."""
    res = []
    if stats is not None:
        assert len(mats) <= MULTI_MAX and tuple(stats.shape) == (len(mats), 2) and stats.dtype == torch.float32 and stats.is_contiguous()
    for lo in range(0, len(mats), MULTI_MAX):
        ms = mats[lo:lo + MULTI_MAX]
        n = len(ms)
        for m in ms:
            assert m.dim() == 2 and m.stride(1) == 1 and m.dtype == torch.float32
        dev = ms[0].device
        device_status(dev)
        # outs = (planes list, stats [n, 2], exps [n]) of an earlier call on the same matrices: overwritten in place (the optimizer
        # keeps writing into these buffers between re-splits: seqcore._weight_planes_h2)
        if outs is not None:
            assert lo == 0 and len(mats) <= MULTI_MAX and stats is None
            os_, st, exps = outs
        else:
            os_ = [Planes.alloc(m.shape[0], m.shape[1], dev, fmt=1) for m in ms]
            st = stats if stats is not None else torch.empty(n, 2, dtype=torch.float32, device=dev)
            exps = torch.empty(n, dtype=torch.int32, device=dev)
        P, I64 = ctypes.c_void_p * n, ctypes.c_int64 * n
        _l.check(_l.load().pxr_h2_split_auto_multi_f32(
            n, P(*[m.data_ptr() for m in ms]), I64(*[m.shape[0] for m in ms]), I64(*[m.shape[1] for m in ms]),
            I64(*[m.stride(0) for m in ms]), P(*[o.ptr().value for o in os_]), I64(*[o.ps for o in os_]), I64(*[o.pr for o in os_]),
            (2 if stats is not None else int(col_stats)) | (int(top) << 8), _l.ptr(st), _l.ptr(exps), _l.stream_ptr()),
            "pxr_h2_split_auto_multi_f32")
        for i, o in enumerate(os_):
            o.exp_dev, o.stats = exps[i:i + 1], st[i]
        res += os_
    if with_buffers:          # (planes, stats [n, 2], exps [n]): what `outs=` takes back to overwrite the same buffers in place
        assert len(mats) <= MULTI_MAX
        return res, st, exps
    return res


ATTN_STAT_SLOTS = 64        # PXR_ATTN_STAT_SLOTS (csrc/attention.hip): words of attn_bwd's `stat` buffer


def ln_bwd_stat_parts(rows: int) -> int:
    """How many partial maxima ln_bwd(stat=...) / bpr_ln_bwd(stat=...) write for `rows` rows (one per workgroup)."""
    return int(_l.load().pxr_ln_bwd_partial_rows(rows))


def split_h2_parts(x: torch.Tensor, parts: torch.Tensor, n_parts: int, bound_with=None) -> Planes:
    """One fp32 matrix -> h2 Planes, its maximum taken from `n_parts` partial maxima its producer left in `parts` (ln_bwd /
    bpr_ln_bwd / attn_bwd with stat=...): no statistics pass, no single-word atomics (pxr_h2_split_parts_f32).
    bound_with = (W Planes, factor): the launch also computes h2_bound_exp(result, W, factor) -- linear_bwd_input_planes(result, W,
    want_planes=True, mul_bound=factor) then finds it on the result and skips its one-thread launch."""
    assert x.dim() == 2 and x.stride(1) == 1 and x.dtype == torch.float32 and parts.dtype == torch.float32 and parts.numel() >= n_parts
    dev = x.device
    device_status(dev)
    o = Planes.alloc(x.shape[0], x.shape[1], dev, fmt=1)
    st = torch.empty(2, dtype=torch.float32, device=dev)
    exps = torch.empty(1, dtype=torch.int32, device=dev)
    bexp = bcol = None
    factor = 1.0
    if bound_with is not None:
        Wp, factor = bound_with
        assert Wp.stats is not None
        bexp = torch.empty(1, dtype=torch.int32, device=dev)
        bcol = ctypes.c_void_p(Wp.stats.data_ptr() + 4)
    _l.check(_l.load().pxr_h2_split_parts_f32(_l.ptr(x), x.shape[0], x.shape[1], x.stride(0), o.ptr(), o.ps, o.pr, _l.ptr(parts),
                                              int(n_parts), _l.ptr(st), _l.ptr(exps), bcol, float(factor), _l.ptr(bexp),
                                              _l.stream_ptr()), "pxr_h2_split_parts_f32")
    o.exp_dev, o.stats = exps, st
    if bound_with is not None:
        o.pre_bound = (bound_with[0].stats.data_ptr(), float(factor), bexp)
    return o


H2_STALE_HEADROOM = 3      # binades between the recent maximum and its placement in the planes (csrc/h2.hip "stale scales")
H2_STALE_DECAY = 0.99      # per-step decay of the running maximum the scales follow (half-life 69 steps)


class H2StaleOverflow(RuntimeError):
    """PXR_STATUS_H2_STALE: a gradient outgrew the headroom of its stale scale and was saturated in one step."""


class H2Sites:
    """Persistent scale state of the gradients a backward pass hands to its GEMMs as h2 planes ("sites": per layer the LayerNorm-2
    backward's output, the LayerNorm-1 backward's output, dqkv).  exps[s] is the exponent site s's producer uses in the NEXT step,
    stats[s] = (max, rows * max) with the headroom applied, bexp[s] the exponent of the planes a GEMM epilogue writes from site s and
    its weight (du).  `update()` -- one launch after the last consumer of a step -- derives all of it from the step's partial maxima
    (pxr_h2_sites_update).  seeded_for: the batch geometry the state was derived from (a different one re-seeds through an exact pass)."""

    def __init__(self, n: int, device):
        assert 1 <= n <= 16
        self.n = n
        self.exps = torch.zeros(n, dtype=torch.int32, device=device)
        self.stats = torch.zeros(n, 2, dtype=torch.float32, device=device)
        self.bexp = torch.zeros(n, dtype=torch.int32, device=device)
        self.run_max = torch.zeros(n, dtype=torch.float32, device=device)
        self.seeded_for = None

    def planes_like(self, s: int, rows: int, cols: int, bound_with=None) -> Planes:
        """Fresh h2 Planes [rows, cols] whose exponent / statistics are site s's (views of the persistent arrays)."""
        o = Planes.alloc(rows, cols, self.exps.device, fmt=1)
        o.exp_dev, o.stats = self.exps[s:s + 1], self.stats[s]
        if bound_with is not None:
            o.pre_bound = (bound_with[0].stats.data_ptr(), float(bound_with[1]), self.bexp[s:s + 1])
        return o

    def update(self, parts, n_parts, rows, bound_b, bound_factor: float):
        """parts[s]: tensor holding site s's partial maxima (first n_parts[s] words); bound_b[s]: None or the weight Planes whose
        column-sum statistic bounds the GEMM output written from site s."""
        n = self.n
        assert len(parts) == len(n_parts) == len(rows) == len(bound_b) == n
        P, I = ctypes.c_void_p * n, ctypes.c_int * n
        bb = [(w.stats.data_ptr() + 4) if w is not None else None for w in bound_b]
        _l.check(_l.load().pxr_h2_sites_update(n, P(*[t.data_ptr() for t in parts]), I(*[int(x) for x in n_parts]), I(*[int(r) for r in rows]),
                                               P(*bb), float(bound_factor), H2_STALE_HEADROOM, H2_STALE_DECAY, _l.ptr(self.run_max),
                                               _l.ptr(self.exps), _l.ptr(self.stats), _l.ptr(self.bexp), _l.stream_ptr()),
                 "pxr_h2_sites_update")


def ln_bwd_h2s(dy, xhat, rstd, gamma, dgamma, dbeta, sites: H2Sites, site: int, stat: torch.Tensor, p_drop=0.0, seed=0, stream_id=0,
               step_dev=None, defer=None, zero: torch.Tensor | None = None, bound_with=None, head=None):
    """A residual LayerNorm site's backward whose GEMM-facing gradient leaves ONLY as h2 planes under the site's stale scale
    (pxr_ln_bwd_h2s_f32), + this step's partial maxima in `stat`.  head = (pos, neg, table, items, masked_index, grad_scale,
    grad_scale_dev): the loss head's backward fused in (dy unused).  -> (dz, Planes, coef | None)."""
    Lb = _l.load()
    D = xhat.shape[-1]
    rows = xhat.numel() // D
    dz = torch.empty_like(xhat)
    ws_bytes = int(Lb.pxr_ln_bwd_ws_bytes(rows, D))
    if defer is not None:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=xhat.device)   # must outlive this call
        defer.add(ws, int(Lb.pxr_ln_bwd_partial_rows(rows)), 2 * D, dgamma, dbeta, D)
        dgamma = dbeta = None
    else:
        ws = _ws.get(ws_bytes, xhat.device)
    assert stat.numel() >= ln_bwd_stat_parts(rows) and (zero is None or zero.numel() <= 256)
    device_status(xhat.device)
    gp = sites.planes_like(site, rows, D, bound_with=bound_with)
    coef = None
    if head is not None:
        pos, neg, table, items, masked_index, grad_scale, grad_scale_dev = head
        B, L = pos.shape
        coef = torch.empty(B, L, dtype=torch.float32, device=pos.device)
        hargs = (_l.ptr(pos), _l.ptr(neg), _l.ptr(table), table.shape[0], _l.ptr(items), _l.ptr(masked_index), B, L, float(grad_scale),
                 _l.ptr(grad_scale_dev), _l.ptr(coef))
    else:
        hargs = (None, None, None, 0, None, None, 0, 0, 1.0, None, None)
    _l.check(Lb.pxr_ln_bwd_h2s_f32(*hargs, _l.ptr(dy), _l.ptr(xhat), _l.ptr(rstd), _l.ptr(gamma), rows, D, _l.ptr(dz), _l.ptr(dgamma),
                                   _l.ptr(dbeta), p_drop, seed, stream_id, _l.ptr(step_dev), _l.ptr(ws), ws_bytes, *_pl(gp),
                                   _l.ptr(gp.exp_dev), _l.ptr(stat), _l.ptr(zero), zero.numel() if zero is not None else 0,
                                   _l.stream_ptr()), "pxr_ln_bwd_h2s_f32")
    return dz, gp, coef


def attn_bwd_h2s(dctx, qkv, probs, B, H, L, d, sites: H2Sites, site: int, stat: torch.Tensor, p_drop=0.0, seed=0, stream_id=0,
                 step_dev=None) -> Planes:
    """dqkv ONLY as h2 Planes [B*L, 3*H*d] under the site's stale scale (pxr_attn_bwd_h2s_f32); `stat`: ATTN_STAT_SLOTS zeroed words."""
    D = H * d
    assert attn_planes_supported(L, d) and (3 * D) % 32 == 0 and stat.numel() >= ATTN_STAT_SLOTS
    _req(dctx, torch.float32, "dctx"); _req(qkv, torch.float32, "qkv"); _req(probs, torch.float32, "probs")
    device_status(qkv.device)
    gp = sites.planes_like(site, B * L, 3 * D)
    cp, base = _l.c_void_p, qkv.data_ptr()
    _l.check(_l.load().pxr_attn_bwd_h2s_f32(_l.ptr(dctx), D, cp(base), cp(base + 4 * D), cp(base + 8 * D), 3 * D, _l.ptr(probs), B, H, L, d,
                                            p_drop, seed, stream_id, _l.ptr(step_dev), *_pl(gp), 3 * D, 0, D, 2 * D, _l.ptr(gp.exp_dev),
                                            _l.ptr(stat), _l.stream_ptr()), "pxr_attn_bwd_h2s_f32")
    return gp


def h2_bound_exp(dy: Planes, W: Planes, factor: float = 1.0) -> torch.Tensor:
    """int32 [1] on the device: the exponent e with |dy W| * factor * 2^e < 2^15 for EVERY element, from max |dy| and the largest
    column sum of |W| (both gathered by split_h2_auto) -- for an input gradient a GEMM epilogue writes as planes before its own
    maximum can be known (pxr_h2_bound_exp)."""
    assert dy.stats is not None and W.stats is not None
    out = torch.empty(1, dtype=torch.int32, device=dy.buf.device)
    _l.check(_l.load().pxr_h2_bound_exp(_l.ptr(dy.stats), ctypes.c_void_p(W.stats.data_ptr() + 4), float(factor), _l.ptr(out),
                                        _l.stream_ptr()), "pxr_h2_bound_exp")
    return out


def split_planes_multi(mats, outs=None, h2: bool = False):
    """Several fp32 matrices -> Planes in ONE launch (pxr_split_planes_multi_f32); `outs`: existing Planes to overwrite.
    h2=True: the two-plane fp16 format, each matrix scaled by its own power of two (h2_exponent of its max |x|: ONE host
    synchronisation for all of them -- meant for weights that are split once)."""
    for m in mats:
        assert m.dim() == 2 and m.stride(1) == 1 and m.dtype == torch.float32
    if h2:
        assert outs is None
        mx = torch.stack([m.abs().max() for m in mats]).tolist()
        outs = [Planes.alloc(m.shape[0], m.shape[1], m.device, fmt=1) for m in mats]
        device_status(mats[0].device)
        for lo in range(0, len(mats), MULTI_MAX):
            ms, os_ = mats[lo:lo + MULTI_MAX], outs[lo:lo + MULTI_MAX]
            n = len(ms)
            for o, v in zip(os_, mx[lo:lo + MULTI_MAX]):
                o.exp = h2_exponent(float(v))
            P, I64, I32 = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
            _l.check(_l.load().pxr_split_h2_multi_f32(
                n, P(*[m.data_ptr() for m in ms]), I64(*[m.shape[0] for m in ms]), I64(*[m.shape[1] for m in ms]),
                I64(*[m.stride(0) for m in ms]), P(*[o.ptr().value for o in os_]), I64(*[o.ps for o in os_]),
                I64(*[o.pr for o in os_]), I32(*[o.exp for o in os_]), _l.stream_ptr()), "pxr_split_h2_multi_f32")
        return outs
    if outs is None:
        outs = [Planes.alloc(m.shape[0], m.shape[1], m.device) for m in mats]
    # the entry point takes at most MULTI_MAX matrices per launch (its argument block is a fixed-size struct): deeper models
    # (4 matrices per layer: n_layers >= 5) take one launch per group
    for lo in range(0, len(mats), MULTI_MAX):
        ms, os_ = mats[lo:lo + MULTI_MAX], outs[lo:lo + MULTI_MAX]
        n = len(ms)
        P, I64 = ctypes.c_void_p * n, ctypes.c_int64 * n
        _l.check(_l.load().pxr_split_planes_multi_f32(
            n, P(*[m.data_ptr() for m in ms]), I64(*[m.shape[0] for m in ms]), I64(*[m.shape[1] for m in ms]),
            I64(*[m.stride(0) for m in ms]), P(*[o.ptr().value for o in os_]), I64(*[o.ps for o in os_]),
            I64(*[o.pr for o in os_]), _l.stream_ptr()), "pxr_split_planes_multi_f32")
    return outs


def linear_fwd_planes(xp: Planes, Wp: Planes, b, gelu: bool = False, save_grad: bool = False, act: str | None = None,
                      want_fp32: bool = True, want_planes: bool = False, lead_shape=None):
    """linear_fwd from planes: y = x W^T + b (xp [M,K], Wp [N,K]).  Returns (y | None, y Planes | None, aux | None) where aux
    is the saved pre-activation / activation derivative of the gelu / act variants (as linear_fwd's second value)."""
    M, K, N = xp.rows, xp.cols, Wp.rows
    dev = xp.buf.device
    shape = tuple(lead_shape) + (N,) if lead_shape is not None else (M, N)
    y = torch.empty(shape, dtype=torch.float32, device=dev) if want_fp32 else None
    yp = Planes.alloc(M, N, dev, fmt=xp.fmt) if want_planes else None
    aux, epi, code = None, EPI_BIAS, 0
    if act is not None and act != "gelu":
        epi, code = EPI_BIAS_ACT_GRAD, ACT_CODES[act]
    elif gelu:
        epi = EPI_BIAS_GELU_GRAD if save_grad else EPI_BIAS_GELU
    if epi != EPI_BIAS:
        aux = torch.empty(shape, dtype=torch.float32, device=dev)
    if xp.fmt:           # fp16 two-plane operands (the output planes, activations, at unit scale)
        gemm_planes(xp, Wp, y.view(M, N) if y is not None else None, epi, bias=b, aux=aux.view(M, N) if aux is not None else None,
                    act=code, Cp=yp)
        return y, yp, aux
    tag = "gemm_p3_kernel<KC,KC,EPI_BIAS_GELU*> (fwd + activation)" if epi != EPI_BIAS else "gemm_p3_kernel<KC,KC,EPI_BIAS> (fwd)"
    with _gemm_timer(2.0 * M * N * K, tag):
        _l.check(_l.load().pxr_gemm_planes_f32(1, M, N, K, xp.ptr(), xp.ps, xp.pr, Wp.ptr(), Wp.ps, Wp.pr, _l.ptr(y), N, epi,
                                               _l.ptr(b), _l.ptr(aux), N, *_pl(yp), code, 0, _l.stream_ptr()),
                 "pxr_gemm_planes_f32")
    return y, yp, aux


def linear_bwd_input_planes(dyp: Planes, Wp: Planes, add: torch.Tensor | None = None, mul: torch.Tensor | None = None,
                            want_fp32: bool = True, want_planes: bool = False, lead_shape=None, mul_bound: float = 1.0):
    """linear_bwd_input from planes: dx = dy W (dyp [M,N], Wp [N,K]) (+ add | * mul).  Returns (dx | None, dx Planes | None).
    h2 operands (both from split_h2_auto): `mul_bound` >= max |mul| enters the bound that fixes the output planes' scale."""
    M, N, K = dyp.rows, dyp.cols, Wp.cols
    assert Wp.rows == N
    dev = dyp.buf.device
    shape = tuple(lead_shape) + (K,) if lead_shape is not None else (M, K)
    dx = torch.empty(shape, dtype=torch.float32, device=dev) if want_fp32 else None
    aux = mul if mul is not None else add
    epi = EPI_MUL if mul is not None else (EPI_ADD if add is not None else EPI_NONE)
    if dyp.fmt:
        # fp16 two-plane operands; the output planes' scale comes from the bound |dy W| * mul_bound (h2_bound_exp)
        dxp = None
        if want_planes:
            dxp = Planes.alloc(M, K, dev, fmt=1)
            pb = dyp.pre_bound
            if pb is not None and Wp.stats is not None and pb[0] == Wp.stats.data_ptr() and pb[1] == float(mul_bound):
                dxp.exp_dev = pb[2]            # the split launch that made dyp computed this bound
            else:
                dxp.exp_dev = h2_bound_exp(dyp, Wp, mul_bound)
        gemm_planes(dyp, Wp, dx.view(M, K) if dx is not None else None, epi, aux=aux.view(M, K) if aux is not None else None, b_kc=False,
                    Cp=dxp)
        return dx, dxp
    dxp = Planes.alloc(M, K, dev) if want_planes else None
    tag = ("gemm_p3_kernel<KC,XC,EPI_MUL> (dX x saved gelu')" if mul is not None else
           "gemm_p3_kernel<KC,XC,EPI_ADD> (dX + residual grad)" if add is not None else "gemm_p3_kernel<KC,XC,EPI_NONE> (dX)")
    with _gemm_timer(2.0 * M * N * K, tag):
        _l.check(_l.load().pxr_gemm_planes_f32(0, M, K, N, dyp.ptr(), dyp.ps, dyp.pr, Wp.ptr(), Wp.ps, Wp.pr, _l.ptr(dx), K,
                                               epi, None, _l.ptr(aux), K, *_pl(dxp), 0, 0, _l.stream_ptr()),
                 "pxr_gemm_planes_f32")
    return dx, dxp


def grouped_dw_planes(problems, tile_hint=0):
    """problems: list of (dy Planes [T,N], x Planes [T,K], dW [N,K] out, db [N] out | None).  One launch for all of them
    (pxr_grouped_dw_planes_f32; one per MULTI_MAX of them: blocks deeper than 4 layers)."""
    if len(problems) > MULTI_MAX:
        for lo in range(0, len(problems), MULTI_MAX):
            grouped_dw_planes(problems[lo:lo + MULTI_MAX], tile_hint)
        return
    L = _l.load()
    n = len(problems)
    P, I, I64 = ctypes.c_void_p * n, ctypes.c_int * n, ctypes.c_int64 * n
    for dy, x, _, _ in problems:
        assert dy.rows == x.rows
    if any(p[0].fmt or p[1].fmt for p in problems):
        if not all(p[0].fmt and p[1].fmt for p in problems):
            raise _l.PxrError("grouped_dw_planes: fp16 two-plane operands need every operand of the launch in that format")
        edev = lambda pl: (pl.exp_dev.data_ptr() if pl.exp_dev is not None else None)
        flops = sum(2.0 * p[0].rows * p[0].cols * p[1].cols for p in problems)
        with _gemm_timer(flops, "grouped_dw_p3_kernel<P4Cfg<..., HALF>> (all dW + db of a block, fp16 two-plane operands)"):
            _l.check(L.pxr_grouped_dw_h2_f32(
                n, P(*[p[0].ptr().value for p in problems]), I64(*[p[0].ps for p in problems]), I64(*[p[0].pr for p in problems]),
                I(*[p[0].exp for p in problems]), P(*[edev(p[0]) for p in problems]),
                P(*[p[1].ptr().value for p in problems]), I64(*[p[1].ps for p in problems]), I64(*[p[1].pr for p in problems]),
                I(*[p[1].exp for p in problems]), P(*[edev(p[1]) for p in problems]),
                P(*[p[2].data_ptr() for p in problems]), P(*[(p[3].data_ptr() if p[3] is not None else None) for p in problems]),
                I(*[p[0].rows for p in problems]), I(*[p[0].cols for p in problems]), I(*[p[1].cols for p in problems]), tile_hint,
                _l.stream_ptr()), "pxr_grouped_dw_h2_f32")
        return
    args = (P(*[p[0].ptr().value for p in problems]), I64(*[p[0].ps for p in problems]), I64(*[p[0].pr for p in problems]),
            P(*[p[1].ptr().value for p in problems]), I64(*[p[1].ps for p in problems]), I64(*[p[1].pr for p in problems]),
            P(*[p[2].data_ptr() for p in problems]), P(*[(p[3].data_ptr() if p[3] is not None else None) for p in problems]),
            I(*[p[0].rows for p in problems]), I(*[p[0].cols for p in problems]), I(*[p[1].cols for p in problems]))
    flops = sum(2.0 * p[0].rows * p[0].cols * p[1].cols for p in problems)
    if tile_hint == 0 and os.environ.get("PXR_P3_DW_TILE"):
        tile_hint = int(os.environ["PXR_P3_DW_TILE"])        # A/B knob (tools/p3_sweep.py)
    with _gemm_timer(flops, "grouped_dw_p3_kernel (all dW + db of the step, from planes)"):
        _l.check(L.pxr_grouped_dw_planes_f32(n, *args, tile_hint, _l.stream_ptr()), "pxr_grouped_dw_planes_f32")


COLSUM_DIRECT_ROWS = int(os.environ.get("PXR_COLSUM_DIRECT_ROWS", "128"))


def colsum(x2d: torch.Tensor, out: torch.Tensor | None = None, defer=None):
    """out[n] = sum_m x[m, n] (deterministic two-stage reduction; second stage deferred when `defer` is given)."""
    L = _l.load()
    _req(x2d, torch.float32, "x")
    M, N = x2d.shape
    o = out if out is not None else torch.empty(N, dtype=torch.float32, device=x2d.device)
    ws_bytes = int(L.pxr_colsum_ws_bytes(M, N))
    if defer is not None and M <= COLSUM_DIRECT_ROWS and x2d.is_contiguous():
        # few rows (the position-embedding gradient of a B = 64 step: 64 rows of L*D): the deferred launch sums the rows themselves --
        # they ARE its partials -- and the first-stage launch is saved
        defer.add(x2d, M, N, o)
        return o
    if defer is not None:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=x2d.device)
        defer.add(ws, int(L.pxr_colsum_partial_rows(M)), N, o)
        _l.check(L.pxr_colsum_f32(_l.ptr(x2d), N, M, N, None, _l.ptr(ws), ws_bytes, _l.stream_ptr()), "pxr_colsum_f32")
        return o
    ws = _ws.get(ws_bytes, x2d.device)
    _l.check(L.pxr_colsum_f32(_l.ptr(x2d), N, M, N, _l.ptr(o), _l.ptr(ws), ws_bytes, _l.stream_ptr()), "pxr_colsum_f32")
    return o


# ------------------------------------------------------------------------------------------------ LayerNorm sites
def _pl(p):
    """(ptr, plane stride, panel rows) of an optional Planes"""
    return (p.ptr(), p.ps, p.pr) if p is not None else (None, 0, 0)


def input_ln_fwd(table, idx, idx_bstride, B, L, pos, gamma, beta, eps, p_drop=0.0, seed=0, stream_id=0, save=True,
                 step_dev=None, planes: bool = False):
    """y = dropout(LN(table[idx[b,t]] + pos[t]))  (sasrec.py:68,77-82 / :99-104).  Returns (y, xhat, rstd); with
    planes=True a 4th value: y as Planes [B*L, D] (written by the same kernel)."""
    Lb = _l.load()
    _req(table, torch.float32, "table"); _req(idx, torch.int64, "idx", contiguous=False)
    N, D = table.shape
    device_status(table.device)
    y = torch.empty(B, L, D, dtype=torch.float32, device=table.device)
    xhat = torch.empty_like(y) if save else None
    rstd = torch.empty(B * L, dtype=torch.float32, device=table.device) if save else None
    # algorithmic bytes: B*L table rows read once + y (+ xhat when saved) written; the [B,2,L+1,D] gather of the
    # reference never exists (SURVEY.md §8d "fused" rule)
    h2 = planes == "h2"
    yp = Planes.alloc(B * L, D, table.device, fmt=int(h2)) if planes else None
    fn = Lb.pxr_input_ln_fwd_h2_f32 if h2 else Lb.pxr_input_ln_fwd_planes_f32
    # (+ the planes of y -- three bf16 ones, 6 B per element, or two fp16 ones -- when the kernel also writes them for the QKV GEMM)
    with _gemm_timer(B * L * D * (4.0 * (3 if save else 2) + ((4.0 if h2 else 6.0) if planes else 0.0)), "ln_fwd_kernel<GATHER> (gather + pos + LN + dropout)"):
        _l.check(fn(_l.ptr(table), N, _l.ptr(idx), idx_bstride, _l.ptr(pos), _l.ptr(gamma),
                                                _l.ptr(beta), eps, B, L, D, _l.ptr(y), _l.ptr(xhat), _l.ptr(rstd), p_drop,
                                                seed, stream_id, _l.ptr(step_dev), *_pl(yp), _l.stream_ptr()),
                 "pxr_input_ln_fwd_f32")
    return (y, xhat, rstd, yp) if planes else (y, xhat, rstd)


def ln_residual_fwd(x, res, gamma, beta, eps, p_drop=0.0, seed=0, stream_id=0, save=True, step_dev=None, planes: bool = False,
                    want_y: bool = True):
    """y = LN(dropout(x) + res)  (layers.py:614-615, :670-671).  Returns (y, xhat, rstd) (+ y as Planes with planes=True;
    want_y=False then skips the fp32 copy of y)."""
    Lb = _l.load()
    _req(x, torch.float32, "x")
    D = x.shape[-1]
    rows = x.numel() // D
    assert want_y or planes
    y = torch.empty_like(x) if want_y else None
    xhat = torch.empty_like(x) if save else None
    rstd = torch.empty(rows, dtype=torch.float32, device=x.device) if save else None
    if planes == "h2":                # y as two fp16 planes (the image tower, the sequence block of large batches)
        yp = Planes.alloc(rows, D, x.device, fmt=1)
        device_status(x.device)
        _l.check(Lb.pxr_ln_residual_fwd_h2_f32(_l.ptr(x), _l.ptr(res), _l.ptr(gamma), _l.ptr(beta), eps, rows, D, _l.ptr(y),
                                               _l.ptr(xhat), _l.ptr(rstd), p_drop, seed, stream_id, _l.ptr(step_dev), *_pl(yp),
                                               _l.stream_ptr()), "pxr_ln_residual_fwd_h2_f32")
        return (y, xhat, rstd, yp)
    yp = Planes.alloc(rows, D, x.device) if planes else None
    _l.check(Lb.pxr_ln_residual_fwd_planes_f32(_l.ptr(x), _l.ptr(res), _l.ptr(gamma), _l.ptr(beta), eps, rows, D, _l.ptr(y),
                                               _l.ptr(xhat), _l.ptr(rstd), p_drop, seed, stream_id, _l.ptr(step_dev),
                                               *_pl(yp), _l.stream_ptr()), "pxr_ln_residual_fwd_f32")
    return (y, xhat, rstd, yp) if planes else (y, xhat, rstd)


def ln_residual_bpr_fwd(x, res, gamma, beta, eps, table, items, masked_index, p_drop=0.0, seed=0, stream_id=0, save=True,
                        step_dev=None):
    """The block's last LayerNorm with the loss head's forward fused in (pxr_ln_residual_bpr_fwd_f32):
    -> (y [B,L,D], xhat, rstd, loss [1], pos [B,L], neg [B,L]) -- what ln_residual_fwd + bpr_loss_fwd return, bit for bit."""
    Lb = _l.load()
    _req(x, torch.float32, "x"); _req(table, torch.float32, "table")
    _req(items, torch.int64, "items"); _req(masked_index, torch.int64, "masked_index")
    B, L, D = x.shape
    dev = x.device
    y = torch.empty_like(x)
    xhat = torch.empty_like(x) if save else None
    rstd = torch.empty(B * L, dtype=torch.float32, device=dev) if save else None
    pos = torch.empty(B, L, dtype=torch.float32, device=dev)
    neg = torch.empty(B, L, dtype=torch.float32, device=dev)
    lossrow = torch.empty(B * L, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    with _gemm_timer(0.0, "ln_fwd_kernel<RESIDUAL + loss head>"):
        _l.check(Lb.pxr_ln_residual_bpr_fwd_f32(_l.ptr(x), _l.ptr(res), _l.ptr(gamma), _l.ptr(beta), eps, B, L, D, _l.ptr(y),
                                                _l.ptr(xhat), _l.ptr(rstd), p_drop, seed, stream_id, _l.ptr(step_dev), _l.ptr(table),
                                                table.shape[0], _l.ptr(items), _l.ptr(masked_index), _l.ptr(pos), _l.ptr(neg),
                                                _l.ptr(lossrow), _l.ptr(loss), _l.stream_ptr()), "pxr_ln_residual_bpr_fwd_f32")
    return y, xhat, rstd, loss, pos, neg


def bpr_ln_bwd(pos, neg, table, items, masked_index, grad_scale, grad_scale_dev, xhat, rstd, gamma, dgamma, dbeta, p_drop=0.0,
               seed=0, stream_id=0, need_dx=False, step_dev=None, defer=None, planes: bool = False, stat: torch.Tensor | None = None):
    """bpr_loss_bwd + ln_bwd(0, ...) of the block's last LayerNorm in one launch (pxr_bpr_ln_bwd_f32): the gradient w.r.t. the
    block's output never reaches HBM.  -> (dz, dx | None, planes of the gradient the next GEMMs read | None, coef [B,L])."""
    Lb = _l.load()
    B, L = pos.shape
    D = xhat.shape[-1]
    rows = B * L
    dz = torch.empty_like(xhat)
    dx = torch.empty_like(xhat) if need_dx else None
    coef = torch.empty(B, L, dtype=torch.float32, device=pos.device)
    ws_bytes = int(Lb.pxr_ln_bwd_ws_bytes(rows, D))
    if defer is not None:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=xhat.device)   # must outlive this call
        defer.add(ws, int(Lb.pxr_ln_bwd_partial_rows(rows)), 2 * D, dgamma, dbeta, D)
        dgamma = dbeta = None
    else:
        ws = _ws.get(ws_bytes, xhat.device)
    assert not (planes and stat is not None)
    gp = Planes.alloc(rows, D, xhat.device) if planes else None
    _l.check(Lb.pxr_bpr_ln_bwd_f32(_l.ptr(pos), _l.ptr(neg), _l.ptr(table), table.shape[0], _l.ptr(items), _l.ptr(masked_index), B, L,
                                   float(grad_scale), _l.ptr(grad_scale_dev), _l.ptr(coef), _l.ptr(xhat), _l.ptr(rstd), _l.ptr(gamma),
                                   D, _l.ptr(dz), _l.ptr(dx), _l.ptr(dgamma), _l.ptr(dbeta), p_drop, seed, stream_id,
                                   _l.ptr(step_dev), _l.ptr(ws), ws_bytes, *_pl(gp), _l.ptr(stat), _l.stream_ptr()),
             "pxr_bpr_ln_bwd_f32")
    return dz, dx, gp, coef


class DeferredReductions:
    """Collects the second stage of several partial reductions (LayerNorm dgamma|dbeta, column sums) so that one
    launch finishes all of them at the end of a backward pass."""

    def __init__(self):
        self.items = []      # (part tensor, P, N, out_a, out_b | None, split)

    def add(self, part, P, N, out_a, out_b=None, split=0):
        self.items.append((part, P, N, out_a, out_b, split))

    def flush(self, bump: torch.Tensor | None = None) -> bool:
        """One launch (per 16 problems).  `bump`: an int64 device counter the LAST launch increments by one; returns
        whether it did (False when there was nothing to reduce: the caller then bumps it itself)."""
        import ctypes

        if not self.items:
            return False
        Lb = _l.load()
        for c0 in range(0, len(self.items), 16):
            it = self.items[c0:c0 + 16]
            n = len(it)
            last = c0 + 16 >= len(self.items)
            Pp, I = ctypes.c_void_p * n, ctypes.c_int * n
            _l.check(Lb.pxr_reduce_partials_multi_f32(
                n, Pp(*[x[0].data_ptr() for x in it]), I(*[x[1] for x in it]), I(*[x[2] for x in it]),
                Pp(*[x[3].data_ptr() for x in it]), Pp(*[(x[4].data_ptr() if x[4] is not None else None) for x in it]),
                I(*[x[5] for x in it]), _l.ptr(bump) if (bump is not None and last) else None, _l.stream_ptr()),
                "pxr_reduce_partials_multi_f32")
        self.items = []
        return bump is not None


def ln_bwd(gather_mode, dy, xhat, rstd, gamma, dgamma, dbeta, p_drop=0.0, seed=0, stream_id=0, need_dx=False,
           step_dev=None, defer: DeferredReductions | None = None, planes: bool = False, stat: torch.Tensor | None = None,
           zero: torch.Tensor | None = None, res: torch.Tensor | None = None):
    """Backward of either LN site; dgamma/dbeta ([D] tensors) are overwritten (by `defer.flush()` when a
    DeferredReductions collector is given).  Returns (dz, dx|None) (+ with planes=True the Planes of dx when it exists,
    else of dz: what the following GEMMs read).  res (a residual site without dropout -- a pre-LN block of the image tower): the
    launch returns dz = res + LayerNorm-backward(dy), and `stat` then holds the partial maxima of that sum (pxr_ln_bwd_res_f32)."""
    Lb = _l.load()
    _req(dy, torch.float32, "dy")
    D = dy.shape[-1]
    rows = dy.numel() // D
    dz = torch.empty_like(dy)
    dx = torch.empty_like(dy) if need_dx else None
    ws_bytes = int(Lb.pxr_ln_bwd_ws_bytes(rows, D))
    if defer is not None:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=dy.device)   # must outlive this call
        defer.add(ws, int(Lb.pxr_ln_bwd_partial_rows(rows)), 2 * D, dgamma, dbeta, D)
        dgamma = dbeta = None
    else:
        ws = _ws.get(ws_bytes, dy.device)
    if res is not None:
        assert not gather_mode and not planes and not need_dx and p_drop == 0.0 and zero is None
        _req(res, torch.float32, "res")
        assert res.numel() == dy.numel() and (stat is None or stat.numel() >= ln_bwd_stat_parts(rows))
        _l.check(Lb.pxr_ln_bwd_res_f32(_l.ptr(dy), _l.ptr(xhat), _l.ptr(rstd), _l.ptr(gamma), _l.ptr(res), rows, D, _l.ptr(dz),
                                       _l.ptr(dgamma), _l.ptr(dbeta), _l.ptr(ws), ws_bytes, _l.ptr(stat), _l.stream_ptr()),
                 "pxr_ln_bwd_res_f32")
        return dz, None
    if stat is not None:        # also: per-workgroup partial maxima of |gradient the next GEMMs read| into stat[:ln_bwd_stat_parts(rows)];
        assert not planes and not gather_mode       # `zero` (<= 256 floats): cleared by the launch (attn_bwd's spread slots)
        assert stat.numel() >= ln_bwd_stat_parts(rows) and (zero is None or zero.numel() <= 256)
        _l.check(Lb.pxr_ln_bwd_stat_f32(_l.ptr(dy), _l.ptr(xhat), _l.ptr(rstd), _l.ptr(gamma), rows, D, _l.ptr(dz), _l.ptr(dx),
                                        _l.ptr(dgamma), _l.ptr(dbeta), p_drop, seed, stream_id, _l.ptr(step_dev), _l.ptr(ws), ws_bytes,
                                        _l.ptr(stat), _l.ptr(zero), zero.numel() if zero is not None else 0, _l.stream_ptr()),
                 "pxr_ln_bwd_stat_f32")
        return dz, dx
    gp = Planes.alloc(rows, D, dy.device) if planes else None
    _l.check(Lb.pxr_ln_bwd_planes_f32(int(gather_mode), _l.ptr(dy), _l.ptr(xhat), _l.ptr(rstd), _l.ptr(gamma), rows, D,
                                      _l.ptr(dz), _l.ptr(dx), _l.ptr(dgamma), _l.ptr(dbeta), p_drop, seed, stream_id,
                                      _l.ptr(step_dev), _l.ptr(ws), ws_bytes, *_pl(gp), _l.stream_ptr()), "pxr_ln_bwd_f32")
    return (dz, dx, gp) if planes else (dz, dx)


# ------------------------------------------------------------------------------------------------ attention
ATTN_FUSED_MAX_L = 128     # sequences beyond this take the batched-GEMM path below (any length)


def _attn_takes_gemm_path(L, d):
    """Beyond 128 positions -- or 65..128 with a head size that is not a multiple of 8, which the two-keys-per-lane fused
    kernels need -- the attention runs as batched GEMMs + the row kernel (any length, head size % 4 == 0)."""
    return L > ATTN_FUSED_MAX_L or (L > 64 and d % 8 != 0)


def _attn_long_fwd(qkv, keymask, km_bstride, B, H, L, d, p_drop, seed, stream_id, step_dev):
    """MAX_ITEM_LIST_LENGTH > 128: S = Q K^T and O = PD V as batched fp32-MFMA GEMMs (grid.z = batch x head) around the
    row kernel pxr_attn_rows_fwd_f32 (mask + softmax + dropout).  Same arithmetic as the fused kernels."""
    Lb = _l.load()
    D, ld, Lp, bh = H * d, 3 * H * d, (L + 3) & ~3, B * H
    P = torch.empty(bh, L, Lp, dtype=torch.float32, device=qkv.device)
    sP, sQ, sC = (H * L * Lp, L * Lp), (L * ld, d), (L * D, d)
    gemm_batched(True, True, L, L, d, qkv, 0, ld, qkv, D, ld, P, 0, Lp, bh, H, sQ, sQ, sP)            # q at 0, k at D
    PD = torch.empty_like(P) if p_drop > 0 else None
    _l.check(Lb.pxr_attn_rows_fwd_f32(_l.ptr(P), _l.ptr(PD), _l.ptr(keymask), km_bstride, B, H, L, Lp, p_drop, seed,
                                      stream_id, _l.ptr(step_dev), d, _l.stream_ptr()), "pxr_attn_rows_fwd_f32")
    ctx = torch.empty(B, L, D, dtype=torch.float32, device=qkv.device)
    gemm_batched(True, False, L, d, L, PD if PD is not None else P, 0, Lp, qkv, 2 * D, ld, ctx, 0, D, bh, H, sP, sQ, sC)
    return ctx, (P, PD)


def _attn_long_bwd(dctx, qkv, saved, B, H, L, d, p_drop, seed, stream_id, step_dev):
    Lb = _l.load()
    P, PD = saved
    D, ld, Lp, bh = H * d, 3 * H * d, (L + 3) & ~3, B * H
    sP, sQ, sC = (H * L * Lp, L * Lp), (L * ld, d), (L * D, d)
    dqkv = torch.empty_like(qkv)
    gemm_batched(False, False, L, d, L, PD if PD is not None else P, 0, Lp, dctx, 0, D, dqkv, 2 * D, ld, bh, H, sP, sC, sQ)  # dV
    dP = torch.empty_like(P)
    gemm_batched(True, True, L, L, d, dctx, 0, D, qkv, 2 * D, ld, dP, 0, Lp, bh, H, sC, sQ, sP)                         # dPD
    _l.check(Lb.pxr_attn_rows_bwd_f32(_l.ptr(P), _l.ptr(dP), B, H, L, Lp, p_drop, seed, stream_id, _l.ptr(step_dev), d,
                                      _l.stream_ptr()), "pxr_attn_rows_bwd_f32")
    gemm_batched(True, False, L, d, L, dP, 0, Lp, qkv, D, ld, dqkv, 0, ld, bh, H, sP, sQ, sQ)                           # dQ = dS K
    gemm_batched(False, False, L, d, L, dP, 0, Lp, qkv, 0, ld, dqkv, D, ld, bh, H, sP, sQ, sQ)                          # dK = dS^T Q
    return dqkv


def attn_planes_supported(L: int, d: int) -> bool:
    return bool(_l.load().pxr_attn_planes_supported(L, d))


def attn_fwd(qkv, keymask, km_bstride, B, H, L, d, p_drop=0.0, seed=0, stream_id=0, save=True, step_dev=None,
             planes: bool = False):
    """qkv [B,L,3*H*d] fused projection output -> (ctx [B,L,H*d], probs [B,H,L,L] | None).  planes=True: ctx is returned
    as Planes [B*L, H*d] INSTEAD of the fp32 tensor (written by the fused kernel where it serves the shape, by a split
    launch otherwise)."""
    Lb = _l.load()
    _req(qkv, torch.float32, "qkv"); _req(keymask, torch.int64, "keymask", contiguous=False)
    D = H * d
    if _attn_takes_gemm_path(L, d):
        ctx, saved = _attn_long_fwd(qkv, keymask, km_bstride, B, H, L, d, p_drop, seed, stream_id, step_dev)
        return (split_planes(ctx.view(B * L, D)) if planes else ctx), (saved if save else None)
    fused_p = planes and attn_planes_supported(L, d)
    h2 = planes == "h2"
    assert fused_p or not h2, "h2 context planes come from the fused kernel only (attn_planes_supported)"
    ctx = None if fused_p else torch.empty(B, L, D, dtype=torch.float32, device=qkv.device)
    cp_ = Planes.alloc(B * L, D, qkv.device, fmt=int(h2)) if fused_p else None
    probs = torch.empty(B, H, L, L, dtype=torch.float32, device=qkv.device) if save else None
    base = qkv.data_ptr()
    q, k, v = _l.c_void_p(base), _l.c_void_p(base + 4 * D), _l.c_void_p(base + 8 * D)
    _l.check((Lb.pxr_attn_fwd_h2_f32 if h2 else Lb.pxr_attn_fwd_planes_f32)(q, k, v, 3 * D, _l.ptr(keymask), km_bstride, B, H, L, d, _l.ptr(ctx), D,
                                        _l.ptr(probs), p_drop, seed, stream_id, _l.ptr(step_dev), *_pl(cp_), _l.stream_ptr()),
             "pxr_attn_fwd_f32")
    if planes and not fused_p:
        cp_ = split_planes(ctx.view(B * L, D))
    return (cp_ if planes else ctx), probs


def attn_bwd(dctx, qkv, probs, B, H, L, d, p_drop=0.0, seed=0, stream_id=0, step_dev=None, planes: bool = False,
             stat: torch.Tensor | None = None):
    """-> dqkv [B,L,3*H*d] laid out like qkv; planes=True: as Planes [B*L, 3*H*d] INSTEAD of the fp32 tensor.  stat (ATTN_STAT_SLOTS
    ZEROED float32 words; shapes of attn_planes_supported only): also max |dqkv|, spread over the words (one atomic per workgroup)."""
    Lb = _l.load()
    D = H * d
    if stat is not None:
        assert not planes and attn_planes_supported(L, d) and stat.numel() >= ATTN_STAT_SLOTS
        _req(dctx, torch.float32, "dctx"); _req(qkv, torch.float32, "qkv"); _req(probs, torch.float32, "probs")
        dqkv = torch.empty_like(qkv)
        cp = _l.c_void_p
        base, g = qkv.data_ptr(), dqkv.data_ptr()
        _l.check(Lb.pxr_attn_bwd_stat_f32(_l.ptr(dctx), D, cp(base), cp(base + 4 * D), cp(base + 8 * D), 3 * D, _l.ptr(probs), B, H, L, d,
                                          cp(g), cp(g + 4 * D), cp(g + 8 * D), 3 * D, p_drop, seed, stream_id, _l.ptr(step_dev),
                                          _l.ptr(stat), _l.stream_ptr()), "pxr_attn_bwd_stat_f32")
        return dqkv
    if _attn_takes_gemm_path(L, d):
        dqkv = _attn_long_bwd(dctx, qkv, probs, B, H, L, d, p_drop, seed, stream_id, step_dev)
        return split_planes(dqkv.view(B * L, 3 * D)) if planes else dqkv
    _req(dctx, torch.float32, "dctx"); _req(qkv, torch.float32, "qkv"); _req(probs, torch.float32, "probs")
    fused_p = planes and attn_planes_supported(L, d) and (3 * D) % 32 == 0
    dqkv = None if fused_p else torch.empty_like(qkv)
    gp = Planes.alloc(B * L, 3 * D, qkv.device) if fused_p else None
    base = qkv.data_ptr()
    cp = _l.c_void_p
    dptr = (lambda o: None) if fused_p else (lambda o: cp(dqkv.data_ptr() + o))
    _l.check(Lb.pxr_attn_bwd_planes_f32(_l.ptr(dctx), D, cp(base), cp(base + 4 * D), cp(base + 8 * D), 3 * D, _l.ptr(probs),
                                        B, H, L, d, dptr(0), dptr(4 * D), dptr(8 * D), 3 * D, p_drop, seed,
                                        stream_id, _l.ptr(step_dev), *_pl(gp), 3 * D, 0, D, 2 * D, _l.stream_ptr()),
             "pxr_attn_bwd_f32")
    if planes and not fused_p:
        gp = split_planes(dqkv.view(B * L, 3 * D))
    return gp if planes else dqkv


# ------------------------------------------------------------------------------------------------ loss head
def bpr_loss_fwd(out, table, items, masked_index):
    """-> (loss [1] on device, pos_score [B,L], neg_score [B,L])   (sasrec.py:88-92)."""
    Lb = _l.load()
    _req(out, torch.float32, "out"); _req(table, torch.float32, "table")
    _req(items, torch.int64, "items"); _req(masked_index, torch.int64, "masked_index")
    B, L, D = out.shape
    dev = out.device
    pos = torch.empty(B, L, dtype=torch.float32, device=dev)
    neg = torch.empty(B, L, dtype=torch.float32, device=dev)
    lossrow = torch.empty(B * L, dtype=torch.float32, device=dev)
    loss = torch.empty(1, dtype=torch.float32, device=dev)
    _l.check(Lb.pxr_bpr_loss_fwd_f32(_l.ptr(out), _l.ptr(table), table.shape[0], _l.ptr(items), _l.ptr(masked_index),
                                     B, L, D, _l.ptr(pos), _l.ptr(neg), _l.ptr(lossrow), _l.ptr(loss),
                                     _l.stream_ptr()), "pxr_bpr_loss_fwd_f32")
    return loss, pos, neg


def bpr_loss_bwd(pos, neg, table, items, masked_index, D, grad_scale=1.0, grad_scale_dev=None):
    """-> (dout [B,L,D], coef [B,L])."""
    Lb = _l.load()
    B, L = pos.shape
    dout = torch.empty(B, L, D, dtype=torch.float32, device=pos.device)
    coef = torch.empty(B, L, dtype=torch.float32, device=pos.device)
    _l.check(Lb.pxr_bpr_loss_bwd_f32(_l.ptr(pos), _l.ptr(neg), _l.ptr(table), table.shape[0], _l.ptr(items),
                                     _l.ptr(masked_index), B, L, D, float(grad_scale), _l.ptr(grad_scale_dev), _l.ptr(dout),
                                     _l.ptr(coef),
                                     _l.stream_ptr()), "pxr_bpr_loss_bwd_f32")
    return dout, coef


# ------------------------------------------------------------------------------------------------ sparse table grad
class SparseRows:
    """(uniq_idx [cap] int64 ascending, rows [cap, D], n [1] int32 on device): the table gradient of one step."""

    def __init__(self, cap: int, D: int, device, packed: bool = False):
        self.cap = cap
        self.packed = None
        if packed:
            # idx | n | rows are views of ONE buffer laid out as pxr.h's packed block, so a data-parallel exchange
            # sends it with a single all-gather (parallel.GradSync) and pxr_merge_packed_rows_f32 reads it as it is
            Lb = _l.load()
            off = int(Lb.pxr_packed_rows_offset(cap))
            buf = torch.zeros(int(Lb.pxr_packed_rows_bytes(cap, D)), dtype=torch.uint8, device=device)
            self.idx = buf[:cap * 8].view(torch.int64)
            self.n = buf[cap * 8:cap * 8 + 4].view(torch.int32)
            self.rows = buf[off:].view(torch.float32).view(cap, D)
            self.packed = buf
            return
        self.idx = torch.zeros(cap, dtype=torch.int64, device=device)
        self.rows = torch.empty(cap, D, dtype=torch.float32, device=device)
        self.n = torch.zeros(1, dtype=torch.int32, device=device)

    def count(self) -> int:  # host sync; tests / logging only
        return int(self.n.item())

    def to_dense(self, n_table: int) -> torch.Tensor:  # tests only
        n = self.count()
        g = torch.zeros(n_table, self.rows.shape[1], dtype=torch.float32, device=self.rows.device)
        keep = self.idx[:n] > 0          # a merged (non-compacted) list marks its empty slots with id 0
        g[self.idx[:n][keep]] = self.rows[:n][keep]
        return g


def embed_grad_rows(idx, rows, n_table, scale=1.0, out: SparseRows | None = None) -> SparseRows:
    """Plain embedding backward in sparse form (see pxr.h)."""
    Lb = _l.load()
    _req(idx, torch.int64, "idx"); _req(rows, torch.float32, "rows")
    n, D = idx.numel(), rows.shape[-1]
    sp = out if out is not None else SparseRows(n, D, rows.device)
    ws_bytes = int(Lb.pxr_embed_grad_ws_bytes(n))
    ws = _ws.get(ws_bytes, rows.device)
    _l.check(Lb.pxr_embed_grad_rows_f32(_l.ptr(idx), n, _l.ptr(rows), D, n_table, float(scale), _l.ptr(sp.idx),
                                        _l.ptr(sp.rows), _l.ptr(sp.n), _l.ptr(ws), ws_bytes, _l.stream_ptr()),
             "pxr_embed_grad_rows_f32")
    return sp


def merge_sorted_rows(idx_all, rows_all, world, n_table, scale=1.0, out: SparseRows | None = None) -> SparseRows:
    """Merge `world` sorted-unique sparse gradients (idx_all [world*cap] padded with ids >= n_table, rows_all
    [world*cap, D]) without re-sorting; the result keeps world*cap slots, empty ones carry id 0 (see pxr.h)."""
    Lb = _l.load()
    _req(idx_all, torch.int64, "idx_all"); _req(rows_all, torch.float32, "rows_all")
    E, D = idx_all.numel(), rows_all.shape[-1]
    cap = E // world
    sp = out if out is not None else SparseRows(E, D, rows_all.device)
    ws_bytes = int(Lb.pxr_merge_rows_ws_bytes(world, cap))
    ws = _ws.get(ws_bytes, rows_all.device)
    _l.check(Lb.pxr_merge_sorted_rows_f32(_l.ptr(idx_all), _l.ptr(rows_all), world, cap, D, n_table, float(scale),
                                          _l.ptr(sp.idx), _l.ptr(sp.rows), _l.ptr(sp.n), _l.ptr(ws), ws_bytes,
                                          _l.stream_ptr()), "pxr_merge_sorted_rows_f32")
    return sp


def merge_packed_rows(packed_all, world, cap, D, n_table, scale=1.0, out: SparseRows | None = None) -> SparseRows:
    """merge_sorted_rows on `world` packed blocks (SparseRows(packed=True).packed of every rank, concatenated by one
    all-gather): counts travel inside the blocks, so the lists need no PAD tail (see pxr.h)."""
    Lb = _l.load()
    _req(packed_all, torch.uint8, "packed_all")
    if packed_all.numel() != world * int(Lb.pxr_packed_rows_bytes(cap, D)):
        raise ValueError("merge_packed_rows: packed_all has the wrong size for (world, cap, D)")
    sp = out if out is not None else SparseRows(world * cap, D, packed_all.device)
    ws_bytes = int(Lb.pxr_merge_rows_ws_bytes(world, cap))
    ws = _ws.get(ws_bytes, packed_all.device)
    _l.check(Lb.pxr_merge_packed_rows_f32(_l.ptr(packed_all), world, cap, D, n_table, float(scale), _l.ptr(sp.idx),
                                          _l.ptr(sp.rows), _l.ptr(sp.n), _l.ptr(ws), ws_bytes, _l.stream_ptr()),
             "pxr_merge_packed_rows_f32")
    return sp


def merge_split_rows(heads_all, rows_all, world, cap, cap_x, D, n_table, scale=1.0, out: SparseRows | None = None):
    """merge on the two-collective exchange with a reduced row capacity (see pxr.h)."""
    Lb = _l.load()
    _req(heads_all, torch.uint8, "heads_all"); _req(rows_all, torch.float32, "rows_all")
    device_status(rows_all.device)
    sp = out if out is not None else SparseRows(world * cap_x, D, rows_all.device)
    ws_bytes = int(Lb.pxr_merge_rows_ws_bytes(world, cap_x))
    ws = _ws.get(ws_bytes, rows_all.device)
    _l.check(Lb.pxr_merge_split_rows_f32(_l.ptr(heads_all), _l.ptr(rows_all), world, cap, cap_x, D, n_table, float(scale),
                                         _l.ptr(sp.idx), _l.ptr(sp.rows), _l.ptr(sp.n), _l.ptr(ws), ws_bytes,
                                         _l.stream_ptr()), "pxr_merge_split_rows_f32")
    return sp


def sample_negatives(pos: torch.Tensor, n_items: int, seed: int, batch_counter: int):
    """pos int64 [B, L+1] left-padded positive windows -> (items [B,2,L+1], masked_index [B,L]) with one negative per
    target position drawn on the device (trainset.py:40-63 semantics; stream selected by (seed, batch_counter))."""
    Lb = _l.load()
    _req(pos, torch.int64, "pos")
    B, W = pos.shape
    items = torch.empty(B, 2, W, dtype=torch.int64, device=pos.device)
    mask = torch.empty(B, W - 1, dtype=torch.int64, device=pos.device)
    _l.check(Lb.pxr_sample_negatives_i64(_l.ptr(pos), B, W, n_items, seed & 0xFFFFFFFFFFFFFFFF,
                                         batch_counter & 0xFFFFFFFFFFFFFFFF, _l.ptr(items), _l.ptr(mask),
                                         _l.stream_ptr()), "pxr_sample_negatives_i64")
    return items, mask


def shard_local_rows(ids: torch.Tensor, world: int, rank: int, n_table: int) -> torch.Tensor:
    """Row-sharded table: local row (id // world + 1) of the ids this rank owns (id % world == rank), 0 elsewhere."""
    Lb = _l.load()
    _req(ids, torch.int64, "ids")
    out = torch.empty_like(ids)
    _l.check(Lb.pxr_shard_local_rows_i64(_l.ptr(ids), ids.numel(), world, rank, n_table, _l.ptr(out), _l.stream_ptr()),
             "pxr_shard_local_rows_i64")
    return out


def shard_bucket_ids(ids: torch.Tensor, n_dev: torch.Tensor, world: int, n_table: int, pp_cap: int, pad_id: int):
    """Ascending unique ids (count on the device) -> per-owner request lists: (req int64 [world, pp_cap], pos int32
    [world, pp_cap], counts int32 [world]) -- pxr_shard_bucket_ids_i64."""
    _req(ids, torch.int64, "ids"); _req(n_dev, torch.int32, "n_dev")
    device_status(ids.device)
    req = torch.empty(world, pp_cap, dtype=torch.int64, device=ids.device)
    pos = torch.empty(world, pp_cap, dtype=torch.int32, device=ids.device)
    counts = torch.empty(world, dtype=torch.int32, device=ids.device)
    _l.check(_l.load().pxr_shard_bucket_ids_i64(_l.ptr(ids), _l.ptr(n_dev), world, n_table, pp_cap, pad_id, _l.ptr(req),
                                                _l.ptr(pos), _l.ptr(counts), _l.stream_ptr()), "pxr_shard_bucket_ids_i64")
    return req, pos, counts


def scatter_rows(src: torch.Tensor, pos: torch.Tensor, dst: torch.Tensor, row_offset: int = 0):
    """dst[row_offset + pos[i], :] = src[i, :] where pos[i] >= 0 (pxr_scatter_rows_f32)."""
    _req(src, torch.float32, "src"); _req(pos, torch.int32, "pos"); _req(dst, torch.float32, "dst")
    D = src.shape[-1]
    assert dst.shape[-1] == D and pos.numel() == src.numel() // D
    _l.check(_l.load().pxr_scatter_rows_f32(_l.ptr(src), _l.ptr(pos), pos.numel(), D, _l.ptr(dst), dst.numel() // D, row_offset,
                                            _l.stream_ptr()), "pxr_scatter_rows_f32")
    return dst


def shard_first_rows(ids_all: torch.Tensor, world: int, rank: int, n_table: int) -> torch.Tensor:
    """shard_local_rows over `world` ascending request lists, keeping each owned id only where it is requested FIRST
    (lowest rank): a duplicate-free work list."""
    Lb = _l.load()
    _req(ids_all, torch.int64, "ids_all")
    out = torch.empty_like(ids_all)
    _l.check(Lb.pxr_shard_first_rows_i64(_l.ptr(ids_all), world, ids_all.numel() // world, rank, n_table, _l.ptr(out),
                                         _l.stream_ptr()), "pxr_shard_first_rows_i64")
    return out


def ids_to_compact(ids: torch.Tensor, uniq_idx: torch.Tensor, n_uniq: torch.Tensor) -> torch.Tensor:
    """1 + position of every id in the ascending unique list (0 for padding): indices into a fetched row block."""
    Lb = _l.load()
    _req(ids, torch.int64, "ids"); _req(uniq_idx, torch.int64, "uniq_idx"); _req(n_uniq, torch.int32, "n_uniq")
    out = torch.empty_like(ids)
    _l.check(Lb.pxr_ids_to_compact_i64(_l.ptr(ids), ids.numel(), _l.ptr(uniq_idx), _l.ptr(n_uniq), _l.ptr(out),
                                       _l.stream_ptr()), "pxr_ids_to_compact_i64")
    return out


def sasrec_embed_grad(items, dx0, out, coef, n_table, scale=1.0, sp: SparseRows | None = None) -> SparseRows:
    Lb = _l.load()
    _req(items, torch.int64, "items"); _req(dx0, torch.float32, "dx0"); _req(out, torch.float32, "out")
    _req(coef, torch.float32, "coef")
    B, L, D = out.shape
    n = 3 * B * L
    sp = sp if sp is not None else SparseRows(n, D, out.device)
    ws_bytes = int(Lb.pxr_embed_grad_ws_bytes(n))
    ws = _ws.get(ws_bytes, out.device)
    _l.check(Lb.pxr_sasrec_embed_grad_f32(_l.ptr(items), B, L, _l.ptr(dx0), _l.ptr(out), _l.ptr(coef), D, n_table,
                                          float(scale), _l.ptr(sp.idx), _l.ptr(sp.rows), _l.ptr(sp.n), _l.ptr(ws),
                                          ws_bytes, _l.stream_ptr()), "pxr_sasrec_embed_grad_f32")
    return sp


def occ_ws_bytes(B: int, L: int) -> int:
    return int(_l.load().pxr_embed_grad_ws_bytes(3 * B * L))


def sasrec_occ_sort(items, n_table, sp: SparseRows, ws: torch.Tensor):
    """Phase 1 (ids only): fills sp.idx / sp.n, leaves the sorted occurrences in `ws` (caller-owned, persistent)."""
    Lb = _l.load()
    _req(items, torch.int64, "items")
    B, _, W = items.shape
    _l.check(Lb.pxr_sasrec_occ_sort(_l.ptr(items), B, W - 1, n_table, _l.ptr(sp.idx), _l.ptr(sp.n), _l.ptr(ws),
                                    ws.numel(), _l.stream_ptr()), "pxr_sasrec_occ_sort")


def occ_split_ws_bytes(B: int, L: int, D: int) -> int:
    """Bytes of the second workspace of sasrec_occ_segsum's split route (0: shape not served).  Allocate it with torch.zeros."""
    return int(_l.load().pxr_sasrec_occ_split_ws_bytes(B, L, D))


def sasrec_occ_segsum(ws: torch.Tensor, dx0, out, coef, n_table, sp: SparseRows, scale=1.0, ws2: torch.Tensor | None = None):
    """Phase 2: sp.rows from the sorted occurrences in `ws`.  ws2 (occ_split_ws_bytes zero-initialised bytes, persistent): very long
    segments are summed by many workgroups (big batches: a popular item's thousands of occurrences)."""
    Lb = _l.load()
    B, L, D = out.shape
    if ws2 is not None:
        _l.check(Lb.pxr_sasrec_occ_segsum_split(_l.ptr(ws), ws.numel(), B, L, _l.ptr(dx0), _l.ptr(out), _l.ptr(coef), D, n_table,
                                                float(scale), _l.ptr(sp.n), _l.ptr(sp.rows), _l.ptr(ws2), ws2.numel(), _l.stream_ptr()),
                 "pxr_sasrec_occ_segsum_split")
        return
    _l.check(Lb.pxr_sasrec_occ_segsum(_l.ptr(ws), ws.numel(), B, L, _l.ptr(dx0), _l.ptr(out), _l.ptr(coef), D,
                                      n_table, float(scale), _l.ptr(sp.n), _l.ptr(sp.rows), _l.stream_ptr()),
             "pxr_sasrec_occ_segsum")


# ------------------------------------------------------------------------------------------------ PixelNet pieces
def mosasrec_emb_grad(dx0, out, coef):
    """-> d_emb [B, L+1, 2, D] (see pxr.h)."""
    Lb = _l.load()
    B, L, D = out.shape
    d = torch.empty(B, L + 1, 2, D, dtype=torch.float32, device=out.device)
    _l.check(Lb.pxr_mosasrec_emb_grad_f32(_l.ptr(dx0), _l.ptr(out), _l.ptr(coef), B, L, D, _l.ptr(d), _l.stream_ptr()),
             "pxr_mosasrec_emb_grad_f32")
    return d


def image_u8_to_f32(store: torch.Tensor, ids: torch.Tensor) -> torch.Tensor:
    """store uint8 [n_store, H, W, 3] on device, ids int64 [...] -> fp32 [..., 3, H, W] normalised like the reference."""
    Lb = _l.load()
    _req(store, torch.uint8, "store"); _req(ids, torch.int64, "ids")
    n_store, H, W, C = store.shape
    if C != 3:
        raise _l.PxrError("image store must be HWC with 3 channels")
    out = torch.empty(*ids.shape, 3, H, W, dtype=torch.float32, device=store.device)
    _l.check(Lb.pxr_image_u8_to_f32(_l.ptr(store), n_store, H, W, _l.ptr(ids), ids.numel(), _l.ptr(out),
                                    _l.stream_ptr()), "pxr_image_u8_to_f32")
    return out


# ------------------------------------------------------------------------------------------------ full-sort eval
def history_csr(history_u: torch.Tensor, history_i: torch.Tensor, B: int, device):
    """(history_u, history_i) of seq_eval_collate (grouped by user, collate_fn.py:27-28) -> (hist_ptr int32 [B+1],
    hist_items int64) on `device`.  Host-side index plumbing (a bincount), not arithmetic."""
    counts = torch.bincount(history_u.cpu(), minlength=B)
    ptr = torch.zeros(B + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(counts, 0).to(torch.int32)
    return ptr.to(device), history_i.to(device=device, dtype=torch.int64).contiguous()


def score_planes_supported(table: torch.Tensor) -> bool:
    """Whether the fused scoring can run its main pass on planes of this table (GEMM mode bf16x3, D % 32 == 0, three planes
    below 2 GiB)."""
    N, D = table.shape
    return gemm_mode() == "bf16x3" and D % 32 == 0 and ((N + 31) // 32 * 32) * D * 6 < 0x7FFFFFF0


def row_norm_max(x: torch.Tensor) -> torch.Tensor:
    """max_i ||x[i]||_2 as a 1-element device tensor (pxr_row_norm_max_f32): the table statistic of the reduced-product top-k."""
    _req(x, torch.float32, "x", contiguous=False)
    assert x.dim() == 2 and x.stride(1) == 1
    out = torch.empty(1, dtype=torch.float32, device=x.device)
    _l.check(_l.load().pxr_row_norm_max_f32(_l.ptr(x), x.shape[0], x.shape[1], x.stride(0), _l.ptr(out), _l.stream_ptr()),
             "pxr_row_norm_max_f32")
    return out


def topk_products() -> int:
    """bf16 products of the top-k threshold pass: PXR_TOPK_PRODUCTS = 6 | 3 | 1 (default 3; results are identical, see pxr.h)."""
    v = int(os.environ.get("PXR_TOPK_PRODUCTS", "3"))
    return v if v in (1, 3, 6) else 3


def score_topk(users: torch.Tensor, ld_users: int, B: int, table: torch.Tensor, K: int, hist_ptr=None,
               hist_items=None, table_planes: Planes | None = None, table_norm_max: torch.Tensor | None = None):
    """Fused full-catalog scoring + masking + top-K (see pxr.h).  `users` may be a strided view (row stride
    ld_users floats).  Returns (topk_idx int64 [B,K], topk_val fp32 [B,K]).  table_planes: split_planes(table), made once
    per evaluation -- the main pass over the catalogue then runs on pre-split operands (pxr_score_topk_planes_f32);
    table_norm_max: row_norm_max(table), made with them -- that pass then runs on topk_products() of the six bf16 products and the
    survivors are re-scored exactly (pxr_score_topk_fast_f32): same ids, same bits."""
    Lb = _l.load()
    _req(users, torch.float32, "users", contiguous=False); _req(table, torch.float32, "table")
    N, D = table.shape
    device_status(table.device)
    idx = torch.empty(B, K, dtype=torch.int64, device=table.device)
    val = torch.empty(B, K, dtype=torch.float32, device=table.device)
    ws_bytes = int(Lb.pxr_score_topk_ws_bytes(B, N, K))
    if ws_bytes < 0:
        raise _l.PxrError("score_topk: K must be in [1, 32]")
    ws = _ws.get(ws_bytes, table.device)
    up = None
    if table_planes is not None:
        u2 = torch.as_strided(users, (B, D), (ld_users, 1))
        up = split_planes(u2)
    products = topk_products() if (table_planes is not None and table_norm_max is not None) else 6
    _l.check(Lb.pxr_score_topk_fast_f32(_l.ptr(users), ld_users, B, _l.ptr(table), N, D, *_pl(up), *_pl(table_planes),
                                        _l.ptr(table_norm_max) if products != 6 else None, products,
                                        _l.ptr(hist_ptr), _l.ptr(hist_items), K, _l.ptr(idx), _l.ptr(val), _l.ptr(ws),
                                        ws_bytes, _l.stream_ptr()), "pxr_score_topk_f32")
    return idx, val


# ------------------------------------------------------------------------------------------------ optimizer
def adamw_flat(p, g, m, v, lr, beta1, beta2, eps, weight_decay, step):
    Lb = _l.load()
    for t, nm in ((p, "p"), (g, "g"), (m, "m"), (v, "v")):
        _req(t, torch.float32, nm)
    _l.check(Lb.pxr_adamw_flat_f32(_l.ptr(p), _l.ptr(g), _l.ptr(m), _l.ptr(v), p.numel(), lr, beta1, beta2, eps,
                                   weight_decay, step, _l.stream_ptr()), "pxr_adamw_flat_f32")


def slot_fill(slot, value=-1):
    Lb = _l.load()
    _req(slot, torch.int32, "slot")
    _l.check(Lb.pxr_slot_fill_i32(_l.ptr(slot), slot.numel(), value, _l.stream_ptr()), "pxr_slot_fill_i32")


def adamw_table(table, m, v, slot, sp: SparseRows | None, lr, beta1, beta2, eps, weight_decay, step):
    Lb = _l.load()
    _req(table, torch.float32, "table"); _req(m, torch.float32, "m"); _req(v, torch.float32, "v")
    _req(slot, torch.int32, "slot")
    N, D = table.shape
    with _gemm_timer(24.0 * N * D + 4.0 * N, "adamw_table_kernel (dense sweep)"):
        _l.check(Lb.pxr_adamw_table_f32(_l.ptr(table), _l.ptr(m), _l.ptr(v), N, D, _l.ptr(slot),
                                        _l.ptr(sp.idx) if sp else None, _l.ptr(sp.rows) if sp else None,
                                        _l.ptr(sp.n) if sp else None, sp.cap if sp else 0, lr, beta1, beta2, eps,
                                        weight_decay, step, _l.stream_ptr()), "pxr_adamw_table_f32")


def adamw_hyper_append(hyper, cumlog, step, lr, beta1, beta2, eps, weight_decay, step_dev=None, advance=False):
    """Append the per-step scalars of `step` (or of *step_dev + 1).  advance=True: end-of-step form -- count the finished
    step on the device first, then append the next step's entry (one launch)."""
    Lb = _l.load()
    _l.check(Lb.pxr_adamw_hyper_append(_l.ptr(hyper), _l.ptr(cumlog), cumlog.numel(), step, _l.ptr(step_dev), lr, beta1,
                                       beta2, eps, weight_decay, int(bool(advance)), _l.stream_ptr()),
             "pxr_adamw_hyper_append")


def adamw_rows(table, m, v, last, hyper, cumlog, t_prev, t_apply, beta1, beta2, eps, rows=None, n_rows=None,
               max_rows=0, grows=None, step_dev=None, step_dev_bias=0, max_blocks=0):
    """Lazy table AdamW: catch rows up through t_prev (+ apply step t_apply with gradient rows).  rows=None: all.
    step_dev (device int64 counter of completed steps) overrides t_prev / t_apply (hipGraph-replayable)."""
    Lb = _l.load()
    N, D = table.shape
    tag = ("adamw_rows_kernel (flush: every row)" if rows is None else
           "adamw_rows_kernel (apply: batch rows + gradient)" if t_apply else
           "adamw_rows_kernel (next batch's rows, beside the GEMMs)" if (step_dev_bias or max_blocks) else
           "adamw_rows_kernel (catch-up: batch rows)")
    with _gemm_timer(0.0, tag):
        _l.check(Lb.pxr_adamw_rows_f32(_l.ptr(table), _l.ptr(m), _l.ptr(v), _l.ptr(last), N, D, _l.ptr(rows),
                                       _l.ptr(n_rows), max_rows, _l.ptr(grows), _l.ptr(hyper), _l.ptr(cumlog), t_prev,
                                       t_apply, _l.ptr(step_dev), int(step_dev_bias), int(max_blocks), beta1, beta2, eps,
                                       _l.stream_ptr()),
                 "pxr_adamw_rows_f32")


def adamw_rows_ids(table, m, v, last, hyper, cumlog, t_prev, beta1, beta2, eps, ids, step_dev=None):
    """Catch-up of the rows named by a raw int64 id tensor (duplicates / 0 / out-of-range allowed): pxr_adamw_rows_ids_f32."""
    _req(ids, torch.int64, "ids")
    N, D = table.shape
    with _gemm_timer(0.0, "adamw_rows_kernel (catch-up: batch rows, claimed from the raw id list)"):
        _l.check(_l.load().pxr_adamw_rows_ids_f32(_l.ptr(table), _l.ptr(m), _l.ptr(v), _l.ptr(last), N, D, _l.ptr(ids),
                                                  ids.numel(), _l.ptr(hyper), _l.ptr(cumlog), t_prev, _l.ptr(step_dev), beta1,
                                                  beta2, eps, _l.stream_ptr()), "pxr_adamw_rows_ids_f32")


def adamw_rows_ids2d(table, m, v, last, hyper, cumlog, t_prev, beta1, beta2, eps, ids, n_lists, row_len, row_stride, step_dev=None,
                     cur_hyper_out=None):
    """adamw_rows_ids over a 2-D window of an int64 id tensor (n_lists rows of row_len ids, row stride row_stride elements):
    pxr_adamw_rows_ids2d_f32.  cur_hyper_out (float32 [4]): also receives the scalars of the step about to run."""
    _req(ids, torch.int64, "ids")
    N, D = table.shape
    assert (n_lists - 1) * row_stride + row_len <= ids.numel()
    with _gemm_timer(0.0, "adamw_rows_kernel (catch-up: input rows, claimed from the raw id window)"):
        _l.check(_l.load().pxr_adamw_rows_ids2d_f32(_l.ptr(table), _l.ptr(m), _l.ptr(v), _l.ptr(last), N, D, _l.ptr(ids), n_lists,
                                                    row_len, row_stride, _l.ptr(hyper), _l.ptr(cumlog), t_prev, _l.ptr(step_dev),
                                                    beta1, beta2, eps, _l.ptr(cur_hyper_out), _l.stream_ptr()),
                 "pxr_adamw_rows_ids2d_f32")


def adamw_flat_tab(p, g, m, v, hyper, step, beta1, beta2, eps, step_dev=None, plane_segments=None, close=None, planes_exps=None):
    """Flat AdamW with the step's scalars from the hyper table.  plane_segments: list of (flat element offset, rows, cols,
    Planes): weight matrices whose updated values are also written as planes by the same launch -- three bf16 planes, or (Planes of
    fmt 1 + planes_exps, an int32 device tensor with one exponent per segment) two fp16 planes scaled by 2^exponent.
    close = (cumlog, cur_hyper, lr, weight_decay): the launch reads this step's scalars from cur_hyper and closes the step itself
    (no adamw_hyper_append(advance=True) launch afterwards)."""
    Lb = _l.load()
    segs = plane_segments or []
    # the fused plane output serves MULTI_MAX matrices; the rest (models with more than 4 layers) are split from the updated flat
    # buffer by a launch of their own right behind the optimizer's
    h2 = bool(segs) and segs[0][3].fmt == 1
    assert not h2 or (planes_exps is not None and planes_exps.dtype == torch.int32 and planes_exps.numel() >= len(segs) and len(segs) <= MULTI_MAX)
    late = segs[MULTI_MAX:]
    segs = segs[:MULTI_MAX]
    n = len(segs)
    if n:
        P, I64 = ctypes.c_void_p * n, ctypes.c_int64 * n
        sa = (n, I64(*[s_[0] for s_ in segs]), I64(*[s_[1] for s_ in segs]), I64(*[s_[2] for s_ in segs]),
              P(*[s_[3].ptr().value for s_ in segs]), I64(*[s_[3].ps for s_ in segs]), I64(*[s_[3].pr for s_ in segs]))
    else:
        sa = (0, None, None, None, None, None, None)
    if close is not None or h2:
        cumlog, cur, lr, wd = close if close is not None else (None, None, 0.0, 0.0)
        with _gemm_timer(0.0, "adamw_flat_tab_kernel" + (" (+ closes the step)" if close is not None else "") + (" (h2 planes out)" if h2 else "")):
            _l.check(Lb.pxr_adamw_flat_tab_ex_f32(_l.ptr(p), _l.ptr(g), _l.ptr(m), _l.ptr(v), p.numel(), _l.ptr(hyper), _l.ptr(cumlog),
                                                  cumlog.numel() if cumlog is not None else 0, step, _l.ptr(step_dev), _l.ptr(cur), lr,
                                                  beta1, beta2, eps, wd, *sa, 1 if h2 else 0, _l.ptr(planes_exps) if h2 else None,
                                                  _l.stream_ptr()), "pxr_adamw_flat_tab_ex_f32")
    else:
        _l.check(Lb.pxr_adamw_flat_tab_planes_f32(_l.ptr(p), _l.ptr(g), _l.ptr(m), _l.ptr(v), p.numel(), _l.ptr(hyper), step,
                                                  _l.ptr(step_dev), beta1, beta2, eps, *sa, _l.stream_ptr()),
                 "pxr_adamw_flat_tab_f32")
    if late:
        split_planes_multi([p[o:o + r * c].view(r, c) for o, r, c, _ in late], [pl for _, _, _, pl in late])


class ScoreClock:
    """The shader clock the fused scoring's main-pass kernels sustain, measured inside them (pxr_score_topk_clock_out): a context
    manager; `.ghz()` after the block has synchronised.  Measurement only (bench.py, tools/eval_bench.py)."""

    def __init__(self, device="cuda"):
        self.buf = torch.zeros(2, dtype=torch.int64, device=device)

    def __enter__(self):
        self.buf.zero_()
        torch.cuda.synchronize()
        _l.check(_l.load().pxr_score_topk_clock_out(_l.ptr(self.buf)), "pxr_score_topk_clock_out")
        return self

    def __exit__(self, *exc):
        torch.cuda.synchronize()
        _l.check(_l.load().pxr_score_topk_clock_out(None), "pxr_score_topk_clock_out")

    def ghz(self) -> float:
        c, r = (int(x) for x in self.buf.tolist())
        return c / r * 0.1 if r > 0 else float("nan")


def counter_add(counter, delta=1):
    Lb = _l.load()
    _req(counter, torch.int64, "counter")
    _l.check(Lb.pxr_counter_add_i64(_l.ptr(counter), delta, _l.stream_ptr()), "pxr_counter_add_i64")
