"""SeqRecCore -- the SASRec sequence encoder + BPR head on the MI355X-native kernels, shared by the ID model
(SASRec, rows come from the item-embedding table) and the pixel model (MOSASRec, rows come from the visual encoder's
output).  Both reference classes run the SAME block (IDNet/sasrec.py:65-92 == PixelNet/mosasrec.py:66-93 after the
item vectors are obtained); here it is one implementation parameterised by the "row source":

    table  [n_rows, D]  fp32   -- the item table, or the flattened encoder output of the batch
    idx    [B, 2, L+1]  int64  -- row ids laid out like SEQTrainDataset's `items` (positives | negatives)

Everything below the class contract is hand-written HIP through the C ABI (pixelrec_amd/ops.py -> include/pxr.h):
  * the nn.Linear / nn.LayerNorm / nn.Embedding sub-modules are PARAMETER CONTAINERS only (reference parameter names
    and init semantics); their forward is never called;
  * all non-table parameters live in ONE flat fp32 buffer (query|key|value weights adjacent => one fused QKV GEMM; one
    flat gradient buffer => one fused AdamW launch and one RCCL all-reduce);
  * gradients are OVERWRITTEN by each backward (the reference calls zero_grad() before every step, trainer.py:117);
  * weight/bias gradients of all layers are computed by ONE grouped GEMM launch, the second stages of all LayerNorm /
    position-embedding reductions by ONE launch.
"""
from __future__ import annotations

import os

import torch
import torch.nn as nn

from .. import ops
from ..lib import PxrError
from .basemodel import BaseModel


# ---- parameter containers with the reference's module tree (names only; their forward is never called) -----
class _MultiHeadAttentionParams(nn.Module):
    def __init__(self, hidden, eps):
        super().__init__()
        self.query = nn.Linear(hidden, hidden)
        self.key = nn.Linear(hidden, hidden)
        self.value = nn.Linear(hidden, hidden)
        self.dense = nn.Linear(hidden, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)


class _FeedForwardParams(nn.Module):
    def __init__(self, hidden, inner, eps):
        super().__init__()
        self.dense_1 = nn.Linear(hidden, inner)
        self.dense_2 = nn.Linear(inner, hidden)
        self.LayerNorm = nn.LayerNorm(hidden, eps=eps)


class _TransformerLayerParams(nn.Module):
    def __init__(self, hidden, inner, eps):
        super().__init__()
        self.multi_head_attention = _MultiHeadAttentionParams(hidden, eps)
        self.feed_forward = _FeedForwardParams(hidden, inner, eps)


class _TransformerEncoderParams(nn.Module):
    def __init__(self, n_layers, hidden, inner, eps):
        super().__init__()
        self.layer = nn.ModuleList([_TransformerLayerParams(hidden, inner, eps) for _ in range(n_layers)])



class SeqRecCore(BaseModel):
    """Parameters + forward/backward of the sequence block; subclasses supply the row source."""

    def _build_core(self, config):
        # hyper-parameters: same config keys as the reference (sasrec.py:16-29 / mosasrec.py:16-28)
        self.n_layers = config["n_layers"]
        self.n_heads = config["n_heads"]
        self.hidden_size = config["embedding_size"]
        self.inner_size = config["inner_size"] * self.hidden_size  # "inner_size" is a multiplier (sasrec.py:21)
        self.hidden_dropout_prob = float(config["hidden_dropout_prob"])
        self.attn_dropout_prob = float(config["attn_dropout_prob"])
        self.hidden_act = config["hidden_act"]
        self.layer_norm_eps = float(config["layer_norm_eps"])
        self.initializer_range = config["initializer_range"]
        self.max_seq_length = config["MAX_ITEM_LIST_LENGTH"]
        if self.hidden_act not in ("gelu", "relu", "swish", "tanh", "sigmoid"):     # ACT2FN, layers.py:642-649
            raise ValueError(f"hidden_act must be one of gelu / relu / swish / tanh / sigmoid, got {self.hidden_act!r}")
        if self.hidden_size % self.n_heads != 0:
            raise ValueError("The hidden size (%d) is not a multiple of the number of attention heads (%d)"
                             % (self.hidden_size, self.n_heads))
        if self.hidden_size % 4 != 0:
            raise ValueError("embedding_size must be a multiple of 4 (16-byte vector accesses)")
        self.position_embedding = nn.Embedding(self.max_seq_length, self.hidden_size)
        self.trm_encoder = _TransformerEncoderParams(self.n_layers, self.hidden_size, self.inner_size,
                                                     self.layer_norm_eps)
        self.LayerNorm = nn.LayerNorm(self.hidden_size, eps=self.layer_norm_eps)
        self.dropout = nn.Dropout(self.hidden_dropout_prob)
        self._init_runtime_state(config)

    def _init_runtime_state(self, config):
        """Everything a model on the flat-buffer / hand-written-backward machinery carries besides its architecture (also used
        by the sibling backbones that bring their own layers: gru4rec.py)."""
        self._flat = None            # packed non-table parameters
        self._gflat = None           # packed gradients (same layout)
        self._views = {}
        self._anchor = None
        self._saved = None
        self.grad_scale = 1.0           # 1/world_size under data parallelism (sum-all-reduce == DDP's mean)
        self._side_stream = None
        self._drop_dev = None              # device counter of completed backward passes (dropout seed offset)
        self.group_weight_grads = True     # all weight/bias gradients of a backward pass in one grouped GEMM launch
        self.overlap_weight_grads = False  # alternative: per-layer launches on a side HIP stream
        # WHERE the grouped weight-gradient launches go (group_weight_grads only):
        #   "grouped"    one launch on the main stream after the input-gradient chain (default);
        #   "fork_layer" one launch per layer on a side stream as soon as the layer's dqkv exists, so that its tiles
        #                co-run with the input-gradient chain of the layers below (a parallel branch of a captured graph);
        #   "fork_half"  two launches per layer (FFN pair after du, attention pair after dqkv): starts earlier;
        #   "fork_tail"  the one grouped launch on a side stream BESIDE the tail of the step (reductions, segmented sum of the
        #                table gradient, the optimizer's row update): memory / latency-bound kernels next to an MFMA-bound
        #                one -- 1.083 -> 1.069 ms replayed from a hipGraph in bf16x3 mode, but +0.14 ms of HOST time per
        #                eagerly issued step (stream waits), which is what a multi-rank step is bound by: not the default.
        #                The per-layer forks below lose outright (1.13 / 1.18 ms).
        # Measured on MI355X (profiles/r02/README.md): the kernels DO overlap (sum of durations 1190 us vs 935 us of
        # union per step) but the aggregate rate does not rise -- both sides slow down, the step gets 2-8 % LONGER
        # (1.199 -> 1.250 / 1.319 ms).  The chip is not short of idle MFMA slots to fill: under sustained fp32-MFMA
        # load it runs at 2.0-2.2 GHz of its 2.4 (tools/diag/gemm_timeline), and more concurrent work lowers the clock.
        self.weight_grad_mode = os.environ.get("PXR_DW_MODE", "grouped")
        # leave the join with the side stream to the first consumer of the flat gradient (wait_flat_grads(): PxrAdamW.step,
        # clip_grad_norm_, GradSync.sync) instead of the end of backward(): the sparse-row update of the optimizer then
        # also runs beside the weight-gradient GEMMs.  Opt-in (GraphedTrainStep / Trainer / bench.py set it): code that
        # reads p.grad right after backward() must not race with the side stream.
        self.defer_weight_grad_join = False
        self._drop_seed = int(config["seed"]) if config["seed"] is not None else 2020
        self._step_counter = 0
        # GEMM operands as pre-split bf16x3 planes (csrc/gemm_p3.cuh): every producer of a GEMM operand (LayerNorm, attention,
        # GEMM epilogues) writes the three bf16 terms once, in the panel layout the GEMM tiles copy linearly, instead of every
        # reading tile splitting fp32 values again.  Same arithmetic as GEMM mode bf16x3 (bit-identical products).  Needs
        # the feature sizes to be multiples of 32; PXR_PLANES=0 turns it off.
        self.use_planes = os.environ.get("PXR_PLANES", "1") != "0"
        self._wplanes = None
        self.register_load_state_dict_post_hook(lambda mod, _keys: mod._after_weights_loaded())

    def _init_weights(self, module):
        """N(0, initializer_range) for every Linear/Embedding weight; LayerNorm (1, 0); biases 0 (sasrec.py:51-61)."""
        if isinstance(module, (nn.Linear, nn.Embedding)):
            module.weight.data.normal_(mean=0.0, std=self.initializer_range)
        elif isinstance(module, nn.LayerNorm):
            module.bias.data.zero_()
            module.weight.data.fill_(1.0)
        if isinstance(module, nn.Linear) and module.bias is not None:
            module.bias.data.zero_()

    # ------------------------------------------------------------------------------------------ flat packing
    def _flat_specs(self):
        specs = [("pos", self.position_embedding.weight), ("ln0.w", self.LayerNorm.weight),
                 ("ln0.b", self.LayerNorm.bias)]
        for i, lay in enumerate(self.trm_encoder.layer):
            a, f = lay.multi_head_attention, lay.feed_forward
            specs += [(f"{i}.q.w", a.query.weight), (f"{i}.k.w", a.key.weight), (f"{i}.v.w", a.value.weight),
                      (f"{i}.q.b", a.query.bias), (f"{i}.k.b", a.key.bias), (f"{i}.v.b", a.value.bias),
                      (f"{i}.o.w", a.dense.weight), (f"{i}.o.b", a.dense.bias),
                      (f"{i}.ln1.w", a.LayerNorm.weight), (f"{i}.ln1.b", a.LayerNorm.bias),
                      (f"{i}.f1.w", f.dense_1.weight), (f"{i}.f1.b", f.dense_1.bias),
                      (f"{i}.f2.w", f.dense_2.weight), (f"{i}.f2.b", f.dense_2.bias),
                      (f"{i}.ln2.w", f.LayerNorm.weight), (f"{i}.ln2.b", f.LayerNorm.bias)]
        return specs

    def _first_flat_parameter(self):
        """The parameter at offset 0 of the flat buffer (its address tells whether the buffer is still the packed one)."""
        return self.position_embedding.weight

    def _ensure_packed(self):
        """(Re)build the flat parameter / gradient buffers when the parameters moved (e.g. after .to(device))."""
        w0 = self._first_flat_parameter()
        if self._flat is not None and self._flat.device == w0.device and w0.data_ptr() == self._flat.data_ptr():
            return
        dev = w0.device
        if dev.type != "cuda":
            raise PxrError("pixelrec_amd models run on a HIP device only (no CPU fallback); move the model with "
                           ".to('cuda') first")
        specs = self._flat_specs()
        total = sum(p.numel() for _, p in specs)
        flat = torch.empty(total, dtype=torch.float32, device=dev)
        gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        off = 0
        views = {}
        for name, p in specs:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            p.grad = gflat[off:off + n].view(p.shape)
            views[name] = (off, n, tuple(p.shape))
            off += n
        self._flat, self._gflat, self._views = flat, gflat, views
        self._wplanes, self._wplanes_fresh = None, False
        self._wplanes_h2, self._wplanes_h2_fresh = None, False
        self._anchor = torch.zeros((), dtype=torch.float32, device=dev, requires_grad=True)
        self._drop_dev = torch.full((1,), self._step_counter, dtype=torch.int64, device=dev)

    def _p(self, name, grad=False, span=1):
        """View of parameter `name` (or of `span` adjacent ones fused along dim 0) in the flat (grad) buffer."""
        off, n, shape = self._views[name]
        buf = self._gflat if grad else self._flat
        if span == 1:
            return buf[off:off + n].view(shape)
        return buf[off:off + span * n].view((span * shape[0],) + tuple(shape[1:]))

    # ---- the dropout "RNG state": masks are a stateless hash of (seed, site, element, step), so the only state is
    # the number of completed backward passes (host mirror + device counter).  Saved / restored with checkpoints.
    def dropout_step(self) -> int:
        return int(self._step_counter)

    def set_dropout_step(self, n: int):
        self._step_counter = int(n)
        if getattr(self, "_drop_dev", None) is not None:
            self._drop_dev.fill_(self._step_counter)

    def flat_parameters(self):
        self._ensure_packed()
        return self._flat, self._gflat

    _flat_grad_waits = ()   # handles of an in-flight all-reduce of the flat gradient (parallel.GradSync, defer_flat)

    _head_out = None        # (loss, pos, neg) left by an _encode that ran the fused loss head
    _dw_join = None         # side stream still computing weight gradients (defer_weight_grad_join)
    _dw_keep = None         # their operands, kept alive until the join is enqueued

    def join_weight_grads(self):
        """Make the current stream wait for weight-gradient GEMMs still running on the side stream (no-op otherwise)."""
        side, self._dw_join = self._dw_join, None
        if side is not None:
            torch.cuda.current_stream().wait_stream(side)
        # the operands of the side-stream launches may be recycled by the allocator only now: whatever reuses their
        # memory on this stream is ordered behind the join
        self._dw_keep = None

    def wait_flat_grads(self):
        """Make the current stream wait for everything that still writes the flat gradient buffer: weight-gradient GEMMs
        on the side stream and a deferred all-reduce (no-op otherwise).  Every consumer of the flat gradient calls this
        first: PxrAdamW.step, clip_grad_norm_."""
        self.join_weight_grads()
        waits, self._flat_grad_waits = self._flat_grad_waits, ()
        for h in waits:
            h.wait()

    def _after_input_ln(self):
        """Hook: runs in the forward right after the input LayerNorm (the first reader of the input rows) has been issued."""
        return None

    def _before_head(self):
        """Hook: runs in the forward between the encoder and the loss head, which is the first reader of the TARGET rows of the
        row source (SASRec joins the side stream that brings those rows up to date beside the encoder)."""
        return None

    def _after_input_grads(self, dx0, coef, saved):
        """Hook: runs inside the backward right after the gradient w.r.t. the gathered rows is known and BEFORE the
        grouped weight-gradient GEMM (so whatever it launches -- e.g. a data-parallel row exchange -- overlaps it)."""
        return None

    # ------------------------------------------------------------------------------------------ planes mode
    def _planes_on(self) -> bool:
        return (self.use_planes and ops.gemm_mode() == "bf16x3" and self.hidden_size % 32 == 0
                and self.inner_size % 32 == 0)

    def _h2_on(self, B: int) -> bool:
        """Whether a forward over B sequences runs its GEMMs on TWO fp16 planes per operand (csrc/planes.cuh "h2": three MFMAs per
        multiply instead of six).  Accuracy: per GEMM of the step no further from fp64 than the f32-input MFMA -- the reference's own
        arithmetic class -- is (0.62-1.02 x its error), and a 40-step AdamW trajectory as close to the f32 mode's as the six-product
        path's: tests/test_gpu_h2.py (round 5), profiles/r05/h2_evidence.log.  The format has a finite range, so every operand that
        is not an O(1) activation carries a power-of-two scale found on the device: the weights' by the optimizer's flat launch
        (which keeps their planes current), the gradients' from partial maxima their producers leave (one split launch per tensor).
        PXR_SEQ_H2 = auto (default) | 1 | 0.  auto = on from PXR_SEQ_H2_MIN_TOKENS tokens per step, default 0: since round 5 the
        scale-finding costs less than the cheaper GEMMs save at every batch size measured on one MI355X (h2 vs six products, ms per
        step: B = 8 0.577 vs 0.605, B = 16 0.547 vs 0.577, B = 32 0.623 vs 0.670, B = 64 0.820 vs 0.916, B = 512 4.20 vs 5.31,
        B = 2048 15.1 vs 19.7; round 4's threshold was 6144 tokens).  PXR_SEQ_H2=0 keeps every GEMM on the six-product bf16x3 planes.
        Read at every forward."""
        mode = os.environ.get("PXR_SEQ_H2", "auto")
        if mode == "0" or not self._planes_on() or not ops.attn_planes_supported(self.max_seq_length, self.hidden_size // self.n_heads):
            return False
        return mode == "1" or B * self.max_seq_length >= int(os.environ.get("PXR_SEQ_H2_MIN_TOKENS", "0"))

    _W_NAMES = (("q.w", 3, "qkv"), ("o.w", 1, "o"), ("f1.w", 1, "f1"), ("f2.w", 1, "f2"))

    _wplanes_fresh = False
    # opt-in (GraphedTrainStep, Trainer, bench.py -- loops in which nothing but the optimizer and load_state_dict touches the
    # weights): trust the planes the optimizer's flat kernel wrote and skip the split launch of the next training forward
    trust_optimizer_planes = False

    def _weight_mats(self):
        return [self._p(f"{i}.{n}", span=sp) for i in range(self.n_layers) for n, sp, _ in self._W_NAMES]

    # bound on max |act'| that scales the du planes (erf-GELU 1.13, the others <= 1.1) x 1.5: the weight statistic the bound uses is as
    # old as the weights' last split (the optimizer keeps their planes current in between)
    _DU_BOUND = 1.13 * 1.5
    _W_TOP = 12                   # weight planes: largest |w| 2^e in [2^11, 2^12) -- four binades (16 x) of room for the optimizer's updates
    _wplanes_h2 = None            # (planes list, stats [n, 2], exps [n]) of the weight matrices as fp16 two-plane operands
    _wplanes_h2_fresh = False     # the optimizer's flat launch wrote them after its update (mark_weight_planes_fresh)
    _planes_fmt_active = 0        # format of the weight planes the LAST training forward used: what the optimizer writes next

    def _weight_planes_h2(self, train: bool = False):
        """The weight matrices as h2 planes.  Scales are found on the device: ONE statistics launch for all of them (max |w| per
        matrix; the input-gradient bound then uses rows x max |w| for the column sums: 2^2-2^3 looser than the true sums, one launch
        instead of eight latency-bound ones) + one split launch -- EXCEPT in a training forward that directly follows an optimizer
        step of PxrAdamW under trust_optimizer_planes: its flat launch wrote the updated weights into these planes itself, with the
        exponents of the last split (weights move by ~lr per step, the exponent leaves two binades of headroom, a value that outgrows
        it raises PXR_STATUS_H2_RANGE; every evaluation / load_state_dict / refresh re-derives the exponents from the values).
        The statistics (max |w|) are then as old as the last split: consumers that bound with them add a safety factor."""
        mats = self._weight_mats()
        keys = [f"{i}.{k}" for i in range(self.n_layers) for _, _, k in self._W_NAMES]
        if not (train and self._wplanes_h2 is not None and self._wplanes_h2_fresh and self.trust_optimizer_planes):
            if len(mats) <= ops.MULTI_MAX:
                self._wplanes_h2 = ops.split_h2_auto(mats, col_stats=False, outs=self._wplanes_h2, with_buffers=True, top=self._W_TOP)
            else:                  # deeper than MULTI_MAX matrices: fresh buffers every forward, nothing for the optimizer to write
                self._wplanes_h2 = None
                self._wplanes_h2_fresh = False
                return dict(zip(keys, ops.split_h2_auto(mats, col_stats=False)))
        self._wplanes_h2_fresh = False
        return dict(zip(keys, self._wplanes_h2[0]))

    def _weight_planes(self, train: bool = False):
        """The block's weight matrices as planes.  Re-split from the flat parameter buffer (ONE launch) at the start of every
        forward -- whoever changed the weights since (load_state_dict, a broadcast, a test poking .data) needs no hook --
        EXCEPT a training-mode forward that directly follows an optimizer step of PxrAdamW, whose flat kernel wrote the
        planes itself (mark_weight_planes_fresh)."""
        mats = self._weight_mats()
        if self._wplanes is None:
            self._wplanes = [ops.Planes.alloc(m.shape[0], m.shape[1], m.device) for m in mats]
            self._wplanes_fresh = False
        if not (train and self._wplanes_fresh and self.trust_optimizer_planes):
            ops.split_planes_multi(mats, self._wplanes)
        self._wplanes_fresh = False
        keys = [f"{i}.{k}" for i in range(self.n_layers) for _, _, k in self._W_NAMES]
        return dict(zip(keys, self._wplanes))

    def weight_plane_segments(self):
        """(flat element offset, rows, cols, Planes) of every weight matrix, for the optimizer's fused plane output -- in the format
        the last training forward used (three bf16 planes, or the fp16 two-plane buffers + their device exponents:
        `weight_plane_exps()`); None when the planes mode is off."""
        if not self._planes_on() or self._flat is None:
            return None
        mats = self._weight_mats()
        base = self._flat.data_ptr()
        if self._planes_fmt_active == 1:
            if self._wplanes_h2 is None:
                return None
            return [((m.data_ptr() - base) // 4, m.shape[0], m.shape[1], pl) for m, pl in zip(mats, self._wplanes_h2[0])]
        if self._wplanes is None:
            self._wplanes = [ops.Planes.alloc(m.shape[0], m.shape[1], m.device) for m in mats]
        return [((m.data_ptr() - base) // 4, m.shape[0], m.shape[1], pl) for m, pl in zip(mats, self._wplanes)]

    def weight_plane_exps(self):
        """int32 device tensor: the exponent of each h2 weight segment (None for the bf16 planes)."""
        return self._wplanes_h2[2] if (self._planes_fmt_active == 1 and self._wplanes_h2 is not None) else None

    def mark_weight_planes_fresh(self):
        if self._planes_fmt_active == 1:
            self._wplanes_h2_fresh = True
        else:
            self._wplanes_fresh = True

    # ---- housekeeping of the planes the optimizer keeps current.  Under trust_optimizer_planes the flat AdamW launch
    # rewrites the h2 weight planes with the exponents (and the consumers bound with the max |w| statistics) of the LAST split.  They
    # are re-derived from the values every PXR_H2_REFRESH_STEPS optimizer steps (default 32: two light launches per 32 steps), so
    # the 2^4 headroom of _W_TOP only has to cover 32 steps of AdamW -- |dw| <= 31.8 lr per step in the worst case, i.e. lr up
    # to 15 max|w| / (32 * 31.8) ~ 1.2e-3 at max|w| = 0.08 -- instead of a whole epoch; and the status word is polled at the same
    # interval without stalling the stream (ops.StatusPoll), so a value that does leave the range raises within two intervals.
    _since_plane_refresh = 0
    _status_poll = None

    def planes_housekeeping(self):
        """Call once per optimizer step from the host loop that trusts the optimizer's planes (GraphedTrainStep.__call__,
        PxrAdamW.step when it is not being captured)."""
        if not self.trust_optimizer_planes or self._flat is None or not self._flat.is_cuda:
            return
        self._since_plane_refresh += 1
        if self._since_plane_refresh < int(os.environ.get("PXR_H2_REFRESH_STEPS", "32")):
            return
        self._since_plane_refresh = 0
        if self._status_poll is None:
            self._status_poll = ops.StatusPoll(self._flat.device)
        self._status_poll.check()
        if self._planes_fmt_active == 1 and self._wplanes_h2 is not None and self._planes_on():
            ops.split_h2_auto(self._weight_mats(), col_stats=False, outs=self._wplanes_h2, top=self._W_TOP)
            self._wplanes_h2_fresh = True
        self._status_poll.start()

    # opt-in like trust_optimizer_planes (GraphedTrainStep, Trainer, bench.py -- loops of similar consecutive steps): the h2 gradient
    # planes of the backward pass are written by their producers under the PREVIOUS step's maxima (ops.H2Sites; PXR_SEQ_H2_STALE=0/1
    # overrides).  A gradient that outgrows its headroom (2^5 x from one step to the next) is saturated and raises PXR_STATUS_H2_STALE.
    h2_stale_scales = False
    _h2_sites = None

    def _h2_stale_on(self) -> bool:
        env = os.environ.get("PXR_SEQ_H2_STALE")
        return (env == "1") or (env != "0" and self.h2_stale_scales)

    def _h2_sites_state(self, device):
        if self._h2_sites is None or self._h2_sites.exps.device != device or self._h2_sites.n != 3 * self.n_layers:
            if 3 * self.n_layers > 16:
                return None
            self._h2_sites = ops.H2Sites(3 * self.n_layers, device)
        return self._h2_sites

    def _after_weights_loaded(self):
        """load_state_dict post hook: new weights mean new gradient magnitudes -- the stale gradient scales are re-seeded by an exact
        pass (a captured step that baked them in must be re-captured by its owner, as after any load) -- and the planes are re-split."""
        if self._h2_sites is not None:
            self._h2_sites.seeded_for = None
            self._h2_sites.run_max.zero_()
        self.refresh_weight_planes()

    def refresh_weight_planes(self):
        """Re-split now (after anything but the optimizer rewrote the weights: load_state_dict, a parameter broadcast, a
        restored snapshot), so that a captured step that trusts the optimizer's planes finds them valid."""
        if self._planes_on() and self._flat is not None and self._wplanes is not None:
            ops.split_planes_multi(self._weight_mats(), self._wplanes)
            self._wplanes_fresh = True
        if self._planes_on() and self._flat is not None and self._wplanes_h2 is not None:
            ops.split_h2_auto(self._weight_mats(), col_stats=False, outs=self._wplanes_h2, top=self._W_TOP)
            self._wplanes_h2_fresh = True

    def _encode_planes(self, table, idx, idx_bstride, B, keymask, km_bstride, train: bool, head=None):
        """_encode with every GEMM operand as planes (same kernels' results, bit for bit: the products are those of GEMM mode
        bf16x3).  fp32 copies exist only where a non-GEMM kernel reads them (residual streams, the attention's qkv)."""
        L, D, H = self.max_seq_length, self.hidden_size, self.n_heads
        d = D // H
        eps = self.layer_norm_eps
        ph = self.hidden_dropout_prob if train else 0.0
        pa = self.attn_dropout_prob if train else 0.0
        seed = (self._drop_seed * 1000003) & 0xFFFFFFFFFFFFFFFF
        sdv = self._drop_dev if train else None
        h2m = self._h2_on(B)
        pf = "h2" if h2m else True        # format of the activation planes the producers write
        saved = {"seed": seed, "ph": ph, "pa": pa, "layers": [], "planes": True, "h2": h2m} if train else None
        wp = self._weight_planes_h2(train) if h2m else self._weight_planes(train)
        if train:
            self._planes_fmt_active = 1 if h2m else 0
        h, xhat0, rstd0, hp = ops.input_ln_fwd(table, idx, idx_bstride, B, L, self._p("pos"), self._p("ln0.w"),
                                               self._p("ln0.b"), eps, ph, seed, 0, save=train, step_dev=sdv, planes=pf)
        self._after_input_ln()
        if train:
            saved["xhat0"], saved["rstd0"], saved["wp"] = xhat0, rstd0, wp
        for i in range(self.n_layers):
            qkv, _, _ = ops.linear_fwd_planes(hp, wp[f"{i}.qkv"], self._p(f"{i}.q.b", span=3), lead_shape=(B, L))
            ctxp, probs = ops.attn_fwd(qkv, keymask, km_bstride, B, H, L, d, pa, seed, 1 + 3 * i, save=train, step_dev=sdv,
                                       planes=pf)
            a, _, _ = ops.linear_fwd_planes(ctxp, wp[f"{i}.o"], self._p(f"{i}.o.b"), lead_shape=(B, L))
            h1, xhat1, rstd1, h1p = ops.ln_residual_fwd(a, h, self._p(f"{i}.ln1.w"), self._p(f"{i}.ln1.b"), eps, ph, seed,
                                                        2 + 3 * i, save=train, step_dev=sdv, planes=pf)
            _, fp, u = ops.linear_fwd_planes(h1p, wp[f"{i}.f1"], self._p(f"{i}.f1.b"), gelu=True, save_grad=train,
                                             act=self.hidden_act, want_fp32=False, want_planes=True, lead_shape=(B, L))
            f2, _, _ = ops.linear_fwd_planes(fp, wp[f"{i}.f2"], self._p(f"{i}.f2.b"), lead_shape=(B, L))
            last = i == self.n_layers - 1
            if last and head is not None:
                # the loss head's forward rides in the block's last LayerNorm launch (ops.ln_residual_bpr_fwd)
                self._before_head()
                h2, xhat2, rstd2, *self._head_out = ops.ln_residual_bpr_fwd(
                    f2, h1, self._p(f"{i}.ln2.w"), self._p(f"{i}.ln2.b"), eps, *head, p_drop=ph, seed=seed, stream_id=3 + 3 * i,
                    save=train, step_dev=sdv)
                r = (h2, xhat2, rstd2, None)
            else:
                r = ops.ln_residual_fwd(f2, h1, self._p(f"{i}.ln2.w"), self._p(f"{i}.ln2.b"), eps, ph, seed, 3 + 3 * i,
                                        save=train, step_dev=sdv, planes=(pf if not last else False))
            h2, xhat2, rstd2 = r[0], r[1], r[2]
            if train:
                saved["layers"].append(dict(h_in=hp, qkv=qkv, probs=probs, ctx=ctxp, xhat1=xhat1, rstd1=rstd1, h1=h1p,
                                            u=u, f=fp, xhat2=xhat2, rstd2=rstd2))
            h = h2
            hp = r[3] if not last else None
        return h, saved

    # ------------------------------------------------------------------------------------------ forward
    # the transformer block fuses the loss head into its last LayerNorm launch (forward) and into that LayerNorm's backward;
    # blocks that bring their own _encode / _backward_core (gru4rec.py, nextitnet.py) set this to False
    _fused_head = True

    def _encode(self, table, idx, idx_bstride, B, keymask, km_bstride, train: bool, head=None):
        """row ids into `table` -> last-layer states [B, L, D] (sasrec.py:68-86 / :97-109); saves activations when
        train.  head = (table, items, masked_index): also run the loss head's forward (results in self._head_out)."""
        if self._planes_on():
            return self._encode_planes(table, idx, idx_bstride, B, keymask, km_bstride, train, head=head)
        L, D, H = self.max_seq_length, self.hidden_size, self.n_heads
        d = D // H
        eps = self.layer_norm_eps
        ph = self.hidden_dropout_prob if train else 0.0
        pa = self.attn_dropout_prob if train else 0.0
        # dropout seed of this step = base + (device counter of completed backward passes): the counter lives on the
        # device so that a captured hipGraph draws fresh masks on every replay
        seed = (self._drop_seed * 1000003) & 0xFFFFFFFFFFFFFFFF
        sdv = self._drop_dev if train else None
        saved = {"seed": seed, "ph": ph, "pa": pa, "layers": []} if train else None
        h, xhat0, rstd0 = ops.input_ln_fwd(table, idx, idx_bstride, B, L, self._p("pos"), self._p("ln0.w"),
                                           self._p("ln0.b"), eps, ph, seed, 0, save=train, step_dev=sdv)
        self._after_input_ln()
        if train:
            saved["xhat0"], saved["rstd0"] = xhat0, rstd0
        for i in range(self.n_layers):
            qkv = ops.linear_fwd(h, self._p(f"{i}.q.w", span=3), self._p(f"{i}.q.b", span=3))
            ctx, probs = ops.attn_fwd(qkv, keymask, km_bstride, B, H, L, d, pa, seed, 1 + 3 * i, save=train, step_dev=sdv)
            a = ops.linear_fwd(ctx, self._p(f"{i}.o.w"), self._p(f"{i}.o.b"))
            h1, xhat1, rstd1 = ops.ln_residual_fwd(a, h, self._p(f"{i}.ln1.w"), self._p(f"{i}.ln1.b"), eps, ph, seed,
                                                   2 + 3 * i, save=train, step_dev=sdv)
            f, u = ops.linear_fwd(h1, self._p(f"{i}.f1.w"), self._p(f"{i}.f1.b"), gelu=True, save_grad=train,
                                  act=self.hidden_act)
            f2 = ops.linear_fwd(f, self._p(f"{i}.f2.w"), self._p(f"{i}.f2.b"))
            if i == self.n_layers - 1 and head is not None:
                self._before_head()
                h2, xhat2, rstd2, *self._head_out = ops.ln_residual_bpr_fwd(
                    f2, h1, self._p(f"{i}.ln2.w"), self._p(f"{i}.ln2.b"), eps, *head, p_drop=ph, seed=seed, stream_id=3 + 3 * i,
                    save=train, step_dev=sdv)
            else:
                h2, xhat2, rstd2 = ops.ln_residual_fwd(f2, h1, self._p(f"{i}.ln2.w"), self._p(f"{i}.ln2.b"), eps, ph, seed,
                                                       3 + 3 * i, save=train, step_dev=sdv)
            if train:
                saved["layers"].append(dict(h_in=h, qkv=qkv, probs=probs, ctx=ctx, xhat1=xhat1, rstd1=rstd1, h1=h1,
                                            u=u, f=f, xhat2=xhat2, rstd2=rstd2))
            h = h2
        return h, saved

    def _forward_core(self, table, items, masked_index, train: bool):
        """loss [1] (device) of the block + BPR head for row ids `items` [B,2,L+1] into `table`; keeps what the
        backward needs in self._saved when train."""
        B = items.shape[0]
        L = self.max_seq_length
        fused = self._fused_head and os.environ.get("PXR_FUSED_HEAD", "1") != "0"
        if fused:
            out, saved = self._encode(table, items, 2 * (L + 1), B, masked_index, L, train=train, head=(table, items, masked_index))
            loss, pos, neg = self._head_out
            self._head_out = None
        else:
            out, saved = self._encode(table, items, 2 * (L + 1), B, masked_index, L, train=train)
            self._before_head()
            loss, pos, neg = ops.bpr_loss_fwd(out, table, items, masked_index)
        if saved is None:  # eval-mode forward (dropout off): activations are not kept, backward is unavailable
            self._saved = None
        else:
            saved.update(out=out, pos=pos, neg=neg, items=items, mask=masked_index, B=B, fused_head=fused)
            self._saved = saved
        self._last_scores = (pos, neg)
        return loss

    # ------------------------------------------------------------------------------------------ backward
    def _backward_core(self, grad_out, table):
        """Backward of _forward_core: fills the flat gradient buffer and returns (dx0 [B,L,D] = gradient w.r.t. the
        gathered input rows, coef [B,L] = d loss / d(pos_score - neg_score), saved dict)."""
        s = self._saved
        if s is None:
            raise PxrError("backward() without a training-mode forward()")
        B, L, D, H = s["B"], self.max_seq_length, self.hidden_size, self.n_heads
        d = D // H
        T = B * L
        seed, ph, pa = s["seed"], s["ph"], s["pa"]
        sdv = self._drop_dev
        g = lambda name, span=1: self._p(name, grad=True, span=span)
        gsd = grad_out.reshape(1).to(torch.float32).contiguous()
        # Weight / bias gradients are off the critical path (only the optimizer consumes them): they are collected
        # and computed by ONE grouped launch at the end (all tiles of all layers in one grid: no split-K, no
        # separate bias reductions), or -- overlap_weight_grads -- per layer on a side stream.
        main = torch.cuda.current_stream()
        side = None
        use_side = self.overlap_weight_grads and not self.group_weight_grads
        if use_side:
            side = self._side_stream
            if side is None or side.device != main.device:
                side = self._side_stream = torch.cuda.Stream(device=main.device)
        pending = []
        fork = self.group_weight_grads and self.weight_grad_mode in ("fork_layer", "fork_half", "fork_tail")
        if fork:
            side = self._side_stream
            if side is None or side.device != main.device:
                side = self._side_stream = torch.cuda.Stream(device=main.device)
        forked = []   # operands of launches already issued on the side stream: kept alive until the join is enqueued

        def fork_pending():
            """Launch the collected (dY, X) pairs on the side stream, behind everything issued so far on `main`."""
            if not pending:
                return
            side.wait_stream(main)
            with torch.cuda.stream(side):
                ops.grouped_linear_bwd_weight(pending)
            forked.extend(pending)
            pending.clear()

        defer = ops.DeferredReductions()   # second stage of every LayerNorm dgamma|dbeta / pos-emb reduction: one launch

        def weight_grads(dy2d, x2d, w_name, b_name, span=1):
            if self.group_weight_grads:
                pending.append((dy2d, x2d, g(w_name, span), g(b_name, span)))
            elif use_side:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    ops.linear_bwd_weight(dy2d, x2d, out=g(w_name, span))
                    ops.colsum(dy2d, out=g(b_name, span))
                dy2d.record_stream(side)
                x2d.record_stream(side)
            else:
                ops.linear_bwd_weight(dy2d, x2d, out=g(w_name, span))
                ops.colsum(dy2d, out=g(b_name, span))

        fused = bool(s.get("fused_head"))
        head_args = (s["pos"], s["neg"], table, s["items"], s["mask"], self.grad_scale, gsd)
        if fused:
            dh = coef = None       # formed inside the last LayerNorm's backward launch (ops.bpr_ln_bwd)
        else:
            dh, coef = ops.bpr_loss_bwd(s["pos"], s["neg"], table, s["items"], s["mask"], D, self.grad_scale, gsd)
        if s.get("planes"):
            # every GEMM operand as planes: the gradients that only GEMMs read (du, dqkv) exist as planes only
            wp = s["wp"]
            pend = []
            h2m = bool(s.get("h2"))

            # h2: every gradient that feeds GEMMs is written as fp32 by its producer, which also leaves PARTIAL maxima of |.| (one per
            # workgroup of a LayerNorm backward; 64 spread words of an attention backward, zeroed by the LayerNorm launch in front of
            # it) in a row of this arena; the split derives the power-of-two scale from them on the device.  Nothing to zero per step.
            n_parts = ops.ln_bwd_stat_parts(T) if h2m else 0
            arena = (torch.empty(3 * self.n_layers, max(n_parts, ops.ATTN_STAT_SLOTS), dtype=torch.float32, device=table.device)
                     if h2m else None)
            slot = iter(range(3 * self.n_layers))
            # stale scales (round 6; csrc/h2.hip): in a loop of similar steps the producers write these planes themselves under the
            # PREVIOUS step's maxima (no split launches, no fp32 copies); the first step of a batch geometry runs the exact path above
            # and seeds the state.  site index = arena row; site_w[s] = the weight behind a site whose GEMM writes planes (du).
            sites = self._h2_sites_state(table.device) if (h2m and self._h2_stale_on()) else None
            geom = (B, L, D, ph > 0)
            stale = sites is not None and sites.seeded_for == geom
            if sites is not None and not stale and sites.seeded_for is not None:
                sites.run_max.zero_()           # another batch geometry: its gradients live on another scale (mean over B)
            site_w = [None] * (3 * self.n_layers)

            def ln_bwd_planes(_mode, dy, xhat, rstd, gamma, dgamma, dbeta, p_drop, seed_, stream_id, zero=None, bound_with=None, **kw):
                """(dz, gradient the next GEMMs read as planes)."""
                if not h2m:
                    dz, _, gp = ops.ln_bwd(0, dy, xhat, rstd, gamma, dgamma, dbeta, p_drop, seed_, stream_id, planes=True, **kw)
                    return dz, gp
                si = next(slot)
                st = arena[si]
                site_w[si] = bound_with[0] if bound_with is not None else None
                if stale:
                    dz, gp, _ = ops.ln_bwd_h2s(dy, xhat, rstd, gamma, dgamma, dbeta, sites, si, st, p_drop, seed_, stream_id,
                                               step_dev=kw.get("step_dev"), defer=kw.get("defer"), zero=zero, bound_with=bound_with)
                    return dz, gp
                dz, dx = ops.ln_bwd(0, dy, xhat, rstd, gamma, dgamma, dbeta, p_drop, seed_, stream_id, stat=st, zero=zero, **kw)
                return dz, ops.split_h2_parts((dx if dx is not None else dz).view(T, D), st, n_parts, bound_with=bound_with)

            def head_ln_bwd_planes(xhat, rstd, gamma, dgamma, dbeta, stream_id, bound_with=None):
                """ln_bwd_planes of the block's last LayerNorm with the loss head's backward fused in: (dz, planes, coef)."""
                kw = dict(p_drop=ph, seed=seed, stream_id=stream_id, need_dx=ph > 0, step_dev=sdv, defer=defer)
                if not h2m:
                    dz, _, gp, cf = ops.bpr_ln_bwd(*head_args, xhat, rstd, gamma, dgamma, dbeta, planes=True, **kw)
                    return dz, gp, cf
                si = next(slot)
                st = arena[si]
                site_w[si] = bound_with[0] if bound_with is not None else None
                if stale:
                    return ops.ln_bwd_h2s(None, xhat, rstd, gamma, dgamma, dbeta, sites, si, st, ph, seed, stream_id, step_dev=sdv,
                                          defer=defer, bound_with=bound_with, head=head_args)
                dz, dx, _, cf = ops.bpr_ln_bwd(*head_args, xhat, rstd, gamma, dgamma, dbeta, stat=st, **kw)
                return dz, ops.split_h2_parts((dx if dx is not None else dz).view(T, D), st, n_parts, bound_with=bound_with), cf

            for i in reversed(range(self.n_layers)):
                a = s["layers"][i]
                # (h2: the split of dxf2 also computes the bound that scales the du planes of the GEMM below)
                bw = (wp[f"{i}.f2"], self._DU_BOUND) if h2m else None
                if fused and i == self.n_layers - 1:
                    dz2, dxf2p, coef = head_ln_bwd_planes(a["xhat2"], a["rstd2"], self._p(f"{i}.ln2.w"), g(f"{i}.ln2.w"),
                                                          g(f"{i}.ln2.b"), 3 + 3 * i, bound_with=bw)
                else:
                    dz2, dxf2p = ln_bwd_planes(0, dh, a["xhat2"], a["rstd2"], self._p(f"{i}.ln2.w"), g(f"{i}.ln2.w"),
                                               g(f"{i}.ln2.b"), ph, seed, 3 + 3 * i, need_dx=ph > 0, step_dev=sdv, defer=defer,
                                               **({"bound_with": bw} if h2m else {}))
                pend.append((dxf2p, a["f"], g(f"{i}.f2.w"), g(f"{i}.f2.b")))
                # (h2: du leaves the epilogue as planes scaled by the bound |dxf2 W2| * max |act'|: erf-GELU 1.13, the others <= 1.1)
                _, dup = ops.linear_bwd_input_planes(dxf2p, wp[f"{i}.f2"], mul=a["u"].view(T, -1), want_fp32=False,
                                                     want_planes=True, mul_bound=self._DU_BOUND)
                pend.append((dup, a["h1"], g(f"{i}.f1.w"), g(f"{i}.f1.b")))
                dh1, _ = ops.linear_bwd_input_planes(dup, wp[f"{i}.f1"], add=dz2.view(T, D), lead_shape=(B, L))
                att_si = next(slot) if h2m else -1
                att_st = arena[att_si][:ops.ATTN_STAT_SLOTS] if h2m else None    # (cleared by the LayerNorm launch below)
                dz1, dxap = ln_bwd_planes(0, dh1, a["xhat1"], a["rstd1"], self._p(f"{i}.ln1.w"), g(f"{i}.ln1.w"),
                                          g(f"{i}.ln1.b"), ph, seed, 2 + 3 * i, need_dx=ph > 0, step_dev=sdv, defer=defer,
                                          **({"zero": att_st} if h2m else {}))
                pend.append((dxap, a["ctx"], g(f"{i}.o.w"), g(f"{i}.o.b")))
                dctx, _ = ops.linear_bwd_input_planes(dxap, wp[f"{i}.o"], lead_shape=(B, L))
                if stale:
                    dqkvp = ops.attn_bwd_h2s(dctx, a["qkv"], a["probs"], B, H, L, d, sites, att_si, att_st, pa, seed, 1 + 3 * i,
                                             step_dev=sdv)
                elif h2m:
                    dqkv = ops.attn_bwd(dctx, a["qkv"], a["probs"], B, H, L, d, pa, seed, 1 + 3 * i, step_dev=sdv, stat=att_st)
                    dqkvp = ops.split_h2_parts(dqkv.view(T, 3 * D), att_st, ops.ATTN_STAT_SLOTS)
                else:
                    dqkvp = ops.attn_bwd(dctx, a["qkv"], a["probs"], B, H, L, d, pa, seed, 1 + 3 * i, step_dev=sdv, planes=True)
                pend.append((dqkvp, a["h_in"], g(f"{i}.q.w", 3), g(f"{i}.q.b", 3)))
                dh, _ = ops.linear_bwd_input_planes(dqkvp, wp[f"{i}.qkv"], add=dz1.view(T, D), lead_shape=(B, L))
            dx0, _ = ops.ln_bwd(1, dh, s["xhat0"], s["rstd0"], self._p("ln0.w"), g("ln0.w"), g("ln0.b"), ph, seed, 0,
                                step_dev=sdv, defer=defer)
            ops.colsum(dx0.view(B, L * D), out=g("pos").view(-1), defer=defer)
            bumped = defer.flush(bump=self._drop_dev)
            # (round 5: "fork_tail" was tried here too -- the grouped launch on a side stream beside the segmented sum and the row
            # update.  Nothing overlaps: the 256x128 ping-pong workgroups hold a CU's whole register file and LDS, so the side
            # branch's kernels queue behind them (profiles/r05/README.md); the planes path keeps one stream.)
            self._after_input_grads(dx0, coef, s)
            ops.grouped_dw_planes(pend)
            if sites is not None:
                # behind the last reader of the site exponents: next step's scales from this step's partial maxima (one launch)
                att = {3 * k + 1 for k in range(self.n_layers)}
                sites.update([arena[q] for q in range(3 * self.n_layers)],
                             [ops.ATTN_STAT_SLOTS if q in att else n_parts for q in range(3 * self.n_layers)],
                             [T] * (3 * self.n_layers), site_w, self._DU_BOUND)
                sites.seeded_for = geom
            self._saved = None
            if not bumped:
                ops.counter_add(self._drop_dev, 1)
            self._step_counter += 1
            return dx0, coef, s
        for i in reversed(range(self.n_layers)):
            a = s["layers"][i]
            # FFN: h2 = LN(dropout(f2) + h1)
            if fused and i == self.n_layers - 1:
                dz2, dxf2, _, coef = ops.bpr_ln_bwd(*head_args, a["xhat2"], a["rstd2"], self._p(f"{i}.ln2.w"), g(f"{i}.ln2.w"),
                                                    g(f"{i}.ln2.b"), p_drop=ph, seed=seed, stream_id=3 + 3 * i, need_dx=ph > 0,
                                                    step_dev=sdv, defer=defer)
            else:
                dz2, dxf2 = ops.ln_bwd(0, dh, a["xhat2"], a["rstd2"], self._p(f"{i}.ln2.w"), g(f"{i}.ln2.w"),
                                       g(f"{i}.ln2.b"), ph, seed, 3 + 3 * i, need_dx=ph > 0, step_dev=sdv, defer=defer)
            if dxf2 is None:
                dxf2 = dz2
            weight_grads(dxf2.view(T, D), a["f"].view(T, -1), f"{i}.f2.w", f"{i}.f2.b")
            du = ops.linear_bwd_input(dxf2, self._p(f"{i}.f2.w"), mul=a["u"])      # a["u"] holds gelu'(pre-activation)
            weight_grads(du.view(T, -1), a["h1"].view(T, D), f"{i}.f1.w", f"{i}.f1.b")
            if fork and self.weight_grad_mode == "fork_half":
                fork_pending()
            dh1 = ops.linear_bwd_input(du, self._p(f"{i}.f1.w"), add=dz2)
            # attention block: h1 = LN(dropout(a) + h)
            dz1, dxa = ops.ln_bwd(0, dh1, a["xhat1"], a["rstd1"], self._p(f"{i}.ln1.w"), g(f"{i}.ln1.w"),
                                  g(f"{i}.ln1.b"), ph, seed, 2 + 3 * i, need_dx=ph > 0, step_dev=sdv, defer=defer)
            if dxa is None:
                dxa = dz1
            weight_grads(dxa.view(T, D), a["ctx"].view(T, D), f"{i}.o.w", f"{i}.o.b")
            dctx = ops.linear_bwd_input(dxa, self._p(f"{i}.o.w"))
            dqkv = ops.attn_bwd(dctx, a["qkv"], a["probs"], B, H, L, d, pa, seed, 1 + 3 * i, step_dev=sdv)
            weight_grads(dqkv.view(T, 3 * D), a["h_in"].view(T, D), f"{i}.q.w", f"{i}.q.b", 3)
            if fork and self.weight_grad_mode != "fork_tail":
                fork_pending()
            dh = ops.linear_bwd_input(dqkv, self._p(f"{i}.q.w", span=3), add=dz1)
        dx0, _ = ops.ln_bwd(1, dh, s["xhat0"], s["rstd0"], self._p("ln0.w"), g("ln0.w"), g("ln0.b"), ph, seed, 0,
                            step_dev=sdv, defer=defer)
        if fork and self.weight_grad_mode == "fork_tail":
            # the ONE grouped weight-gradient launch goes to the side stream here; the main stream carries on with the
            # small reductions, the segmented sum of the table gradient and (defer_weight_grad_join) the optimizer's
            # row update -- memory/latency-bound kernels beside an MFMA-bound one
            fork_pending()
        ops.colsum(dx0.view(B, L * D), out=g("pos").view(-1), defer=defer)
        # every dropout-mask consumer of this pass has been issued: the reduction launch also advances the dropout
        # step counter (saves a 1-thread launch per step)
        bumped = defer.flush(bump=self._drop_dev)
        self._after_input_grads(dx0, coef, s)   # model-specific tail that only needs dx0/coef (table rows, exchange)
        if pending:
            ops.grouped_linear_bwd_weight(pending)
        if use_side:
            main.wait_stream(side)
        if fork:
            self._dw_join, self._dw_keep = side, forked
            if not self.defer_weight_grad_join:
                self.join_weight_grads()
        self._saved = None
        if not bumped:
            ops.counter_add(self._drop_dev, 1)
        self._step_counter += 1
        return dx0, coef, s

