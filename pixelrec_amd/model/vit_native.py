"""The PixelNet image encoder on the MI355X-native kernels: forward of every CLIP ViT block, backward of the trainable
ones, the MeanItemEncoder head, and the flat AdamW state of the 'visual_encoder' parameter group.

Reference: HF `CLIPVisionModel` as built by code/REC/model/load.py:90-120 (first `tune_scale` named parameters frozen,
`post_layernorm` -> Identity) wrapped by `MeanItemEncoder` / the cls variant (code/REC/model/layers.py:65-128), trained
by the 'visual_encoder' group of the reference optimizer (code/REC/trainer/trainer.py:74-96).

One pre-LN block  x -> x + out_proj(attn(LN1 x)) -> (.) + fc2(quick_gelu(fc1(LN2 .)))  is, on the device:
    LN1                          pxr_ln_residual_fwd_f32 (no residual, eps 1e-5)
    k|v|q projection             ONE fp32-MFMA GEMM over the fused [3H, H] weight (+bias)
    S = Q K^T, P = softmax, P V  two BATCHED GEMM launches (grid.z = image x head) + pxr_softmax_rows_f32
    out_proj + bias + residual   GEMM, EPI_BIAS_ADD
    LN2, fc1 + quick_gelu        GEMM, EPI_BIAS_QGELU_GRAD (saves quick_gelu' for the backward)
    fc2 + bias + residual        GEMM, EPI_BIAS_ADD
and the backward mirrors it with the input-gradient GEMMs (x gelu' / + residual-gradient epilogues), four batched
attention contractions, pxr_softmax_rows_bwd_f32, pxr_ln_bwd_f32 and ONE grouped weight/bias-gradient launch per block.
Nothing here is a torch op except tensor allocation and the im2col view/copy of the pixel tensor (a layout change).

All encoder parameters live in ONE flat fp32 buffer (`nn.Parameter.data` are views; k|v|q weights adjacent so the fused
projection needs no copy); gradients of the trainable ones are views of a flat gradient buffer that every backward
overwrites, and `VisualAdamW` updates the trainable segments with the same fused kernel as the rest of the model.
"""
from __future__ import annotations

import os

import torch

from .. import ops
from ..lib import PxrError

# intra-block flat order: the three projection weights adjacent, then their biases (registration order is k, v, q)
_BLOCK_ORDER = ("self_attn.k_proj.weight", "self_attn.v_proj.weight", "self_attn.q_proj.weight",
                "self_attn.k_proj.bias", "self_attn.v_proj.bias", "self_attn.q_proj.bias",
                "self_attn.out_proj.weight", "self_attn.out_proj.bias", "layer_norm1.weight", "layer_norm1.bias",
                "mlp.fc1.weight", "mlp.fc1.bias", "mlp.fc2.weight", "mlp.fc2.bias", "layer_norm2.weight",
                "layer_norm2.bias")


class NativeTower:
    """Flat packing + native forward/backward of a visual._ItemEncoderBase (`enc`)."""

    def __init__(self, enc):
        self.enc = enc
        self.flat = self.gflat = None
        self.views = {}          # name (relative to enc) -> (offset, numel, shape)
        self.segments = []       # [(lo, hi)] flat ranges that receive gradients and optimizer updates
        self._scratch = {}
        self._wplanes = {}       # block index -> weight planes of a forward-only block (planes mode)

    # ------------------------------------------------------------------------------------------ packing
    def _ordered(self):
        e = self.enc
        named = dict(e.named_parameters())
        vm = "item_encoder.vision_model."
        order = [vm + "embeddings.class_embedding", vm + "embeddings.patch_embedding.weight",
                 vm + "embeddings.position_embedding.weight", vm + "pre_layrnorm.weight", vm + "pre_layrnorm.bias"]
        n_layers = len(e.item_encoder.vision_model.encoder.layers)
        for i in range(n_layers):
            order += [f"{vm}encoder.layers.{i}.{s}" for s in _BLOCK_ORDER]
        # post_layernorm is Identity in a load_model() encoder (load.py:112,116); a bare CLIPVisionEncoder still carries it
        order += [k for k in (vm + "post_layernorm.weight", vm + "post_layernorm.bias") if k in named]
        for wn, bn, _ in e.head_layers():       # one Linear, or the layers of the `dnn_layers` MLP head (layers.py:69-71)
            order += [wn, bn]
        assert set(order) == set(named), sorted(set(named) ^ set(order))
        return [(k, named[k]) for k in order]

    def ensure_packed(self):
        first = self.enc.item_encoder.vision_model.embeddings.class_embedding
        if self.flat is not None and self.flat.device == first.device and first.data_ptr() == self.flat.data_ptr():
            return
        dev = first.device
        self._require_hip(dev)
        specs = self._ordered()
        pad4 = lambda n: (n + 3) & ~3                       # every tensor starts 16-byte aligned
        total = sum(pad4(p.numel()) for _, p in specs)
        flat = torch.zeros(total, dtype=torch.float32, device=dev)
        gflat = torch.zeros(total, dtype=torch.float32, device=dev)
        off, views, segs = 0, {}, []
        for name, p in specs:
            n = p.numel()
            flat[off:off + n].copy_(p.data.reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            views[name] = (off, n, tuple(p.shape))
            # post_layernorm: unused by 'mean' / 'cls' (load.py:112), the pooled head's LayerNorm under 'pool'
            if p.requires_grad and ("post_layernorm" not in name or self.enc.native_method == "pool"):
                p.grad = gflat[off:off + n].view(p.shape)
                if segs and segs[-1][1] == off:
                    segs[-1] = (segs[-1][0], off + pad4(n))
                else:
                    segs.append((off, off + pad4(n)))
            off += pad4(n)
        self.flat, self.gflat, self.views, self.segments = flat, gflat, views, segs
        self._wplanes = {}

    @staticmethod
    def _require_hip(dev):
        if dev.type != "cuda":
            raise PxrError("the native image encoder runs on a HIP device only (no CPU fallback)")

    def view(self, name, grad=False, span=1):
        off, n, shape = self.views[name]
        buf = self.gflat if grad else self.flat
        if span == 1:
            return buf[off:off + n].view(shape)
        return buf[off:off + span * n].view((span * shape[0],) + tuple(shape[1:]))

    def first_trainable_block(self):
        """Index of the first block holding a trainable parameter (n_layers if none); -1 if the embeddings train."""
        e = self.enc
        vm = e.item_encoder.vision_model
        emb = [vm.embeddings.class_embedding, vm.embeddings.patch_embedding.weight, vm.embeddings.position_embedding.weight,
               vm.pre_layrnorm.weight, vm.pre_layrnorm.bias]
        if any(p.requires_grad for p in emb):
            return -1
        for i, layer in enumerate(vm.encoder.layers):
            if any(p.requires_grad for p in layer.parameters()):
                return i
        return len(vm.encoder.layers)

    # ------------------------------------------------------------------------------------------ forward
    def _shape(self):
        vm = self.enc.item_encoder.vision_model
        H = vm.embeddings.class_embedding.shape[0]
        heads = vm.encoder.layers[0].self_attn.heads
        return H, heads, H // heads, vm.embeddings.num_patches + 1, vm.embeddings.patch_size

    def _attn_fwd(self, qkv, n, T, heads, d, keep):
        H, ld, Tp = heads * d, 3 * heads * d, (T + 3) & ~3
        if not keep and ops.tower_attn_supported(T, d):
            return ops.tower_attn_fwd(qkv.view(n * T, ld), n, T, heads, d, 2 * H, 0, H, d ** -0.5)[0], None
        key = ("S", n, T)
        S = None if keep else self._scratch.get(key)
        if S is None:
            S = torch.empty(n * heads, T, Tp, dtype=torch.float32, device=qkv.device)
            if not keep:
                self._scratch = {key: S}
        bh = n * heads
        # S = Q K^T (q rows at 2H + h*d, k rows at h*d of the fused k|v|q projection)
        ops.gemm_batched(True, True, T, T, d, qkv, 2 * H, ld, qkv, 0, ld, S, 0, Tp, bh, heads, (T * ld, d), (T * ld, d),
                         (heads * T * Tp, T * Tp))
        ops.softmax_rows(S, bh * T, T, Tp, d ** -0.5)
        ctx = torch.empty(n, T, H, dtype=torch.float32, device=qkv.device)
        # O = P V (v rows at H + h*d)
        ops.gemm_batched(True, False, T, d, T, S, 0, Tp, qkv, H, ld, ctx, 0, H, bh, heads, (heads * T * Tp, T * Tp),
                         (T * ld, d), (T * H, d))
        return ctx, (S if keep else None)

    # ---- planes mode (csrc/gemm_p3.cuh): the blocks that only run FORWARD (the frozen front of the tower: 10 of 12 blocks at
    # the shipped tune_scale, every block at inference) take their GEMM operands as pre-split planes -- LayerNorm, the fused
    # attention and the fc1 epilogue write them, the frozen weights are split once.  FROZEN blocks: two fp16 planes, three
    # products per multiply (_h2_block); the others: three bf16 planes, the same products as GEMM mode bf16x3, bit for bit.
    def _planes_on(self):
        H, heads, d, T, _ = self._shape()
        return (os.environ.get("PXR_PLANES", "1") != "0" and ops.gemm_mode() == "bf16x3" and H % 32 == 0
                and self.view(f"item_encoder.vision_model.encoder.layers.0.mlp.fc1.weight").shape[0] % 32 == 0)

    def _h2_block(self, i, train=False):
        """Whether block i runs on the TWO-plane fp16 operands (csrc/planes.cuh "h2": three MFMAs per multiply instead of six, the
        same 2^-22-grade accuracy -- profiles/r04/lab/h2_lab_run1.log), when the fused attention serves the shape.  Frozen blocks
        (weights split once, scales chosen on the host): PXR_TOWER_H2 (default on).  Blocks whose weights move (forward AND
        backward of a training step, `train`; their inference forwards too): PXR_TOWER_H2_TRAIN (default on) -- every scale is
        then chosen on the device (ops.split_h2_auto / ops.h2_bound_exp)."""
        H, heads, d, T, _ = self._shape()
        if not ops.tower_attn_supported(T, d) or os.environ.get("PXR_TOWER_H2", "1") == "0":
            return False
        if i < self.first_trainable_block() and not train:
            return True
        return os.environ.get("PXR_TOWER_H2_TRAIN", "1") != "0" and os.environ.get("PXR_TOWER_ATTN_BWD", "1") != "0"

    def _block_weight_planes(self, i, h2=False):
        """(qkv, out_proj, fc1, fc2) planes of block i: one split launch; kept for frozen blocks (ensure_packed /
        load_state_dict drop the cache), redone at every forward for trainable ones."""
        wp = self._wplanes.get((i, h2))
        if wp is None:
            vm = "item_encoder.vision_model."
            P = lambda s_, **kw: self.view(f"{vm}encoder.layers.{i}.{s_}", **kw)
            mats = [P("self_attn.k_proj.weight", span=3), P("self_attn.out_proj.weight"), P("mlp.fc1.weight"), P("mlp.fc2.weight")]
            frozen = i < self.first_trainable_block()
            if h2 and not frozen:
                wp = ops.split_h2_auto(mats, col_stats=True)      # scales on the device: no host synchronisation per step
            else:
                wp = ops.split_planes_multi(mats, h2=h2)
            if frozen:
                self._wplanes[(i, h2)] = wp      # only FROZEN blocks are cached: a trainable block's weights move every step
        return wp

    def drop_weight_planes(self, trainable_only: bool = False):
        """Forget cached weight planes: all of them (weights reloaded), or those of the blocks an optimizer step just rewrote
        (they are only ever used by inference forwards between training steps)."""
        if trainable_only:
            first = self.first_trainable_block()
            self._wplanes = {k: v for k, v in self._wplanes.items() if k[0] < first}
        else:
            self._wplanes = {}

    def _block_fwd_planes(self, i, x, keep=False, out_planes=False):
        """One block on the planes GEMMs.  keep=True (a trainable block of a training step) also returns what its backward
        needs -- activations as planes where the weight-gradient / input-gradient GEMMs read them."""
        vm = "item_encoder.vision_model."
        P = lambda s_, **kw: self.view(f"{vm}encoder.layers.{i}.{s_}", **kw)
        H, heads, d, T, _ = self._shape()
        n = x.shape[0]
        M = n * T
        h2 = self._h2_block(i, train=keep)
        pf = "h2" if h2 else True          # format of the activation planes this block's producers write
        Wqkv, Wo, W1, W2 = W = self._block_weight_planes(i, h2)
        _, xh1, rs1, h1p = ops.ln_residual_fwd(x, None, P("layer_norm1.weight"), P("layer_norm1.bias"), 1e-5, save=keep, planes=pf, want_y=False)
        qkv = torch.empty(n, T, 3 * H, dtype=torch.float32, device=x.device)
        ops.gemm_planes(h1p, Wqkv, qkv.view(M, 3 * H), ops.EPI_BIAS, bias=P("self_attn.k_proj.bias", span=3))
        S = ctx = lse = None
        fused_bwd = keep and ops.tower_attn_supported(T, d) and os.environ.get("PXR_TOWER_ATTN_BWD", "1") != "0"
        if (not keep or fused_bwd) and ops.tower_attn_supported(T, d):
            # fused: no score matrix, the context leaves as planes; a trainable block also keeps the fp32 context and the
            # log-sum-exp its fused backward recomputes the probabilities from
            ctx, ctxp, lse = ops.tower_attn_fwd(qkv.view(M, 3 * H), n, T, heads, d, 2 * H, 0, H, d ** -0.5, ctx=fused_bwd,
                                                planes=pf, lse=fused_bwd)
        else:
            ctx, S = self._attn_fwd(qkv, n, T, heads, d, keep)
            ctxp = ops.split_planes(ctx.view(M, H))
        x2 = torch.empty_like(x)
        ops.gemm_planes(ctxp, Wo, x2.view(M, H), ops.EPI_BIAS_ADD, bias=P("self_attn.out_proj.bias"), aux=x.view(M, H))
        _, xh2, rs2, h2p = ops.ln_residual_fwd(x2, None, P("layer_norm2.weight"), P("layer_norm2.bias"), 1e-5, save=keep, planes=pf, want_y=False)
        fp = ops.Planes.alloc(M, W1.rows, x.device, fmt=int(h2))
        gq = torch.empty(M, W1.rows, dtype=torch.float32, device=x.device) if keep else None
        ops.gemm_planes(h2p, W1, None, ops.EPI_BIAS_QGELU_GRAD if keep else ops.EPI_BIAS_QGELU, bias=P("mlp.fc1.bias"), aux=gq, Cp=fp)
        x3 = torch.empty_like(x)
        # out_planes (the LAST block when the head's Linear runs on h2 operands): its output also leaves as h2 planes, unit scale
        x3p = ops.Planes.alloc(M, H, x.device, fmt=1) if (out_planes and h2) else None
        ops.gemm_planes(fp, W2, x3.view(M, H), ops.EPI_BIAS_ADD, bias=P("mlp.fc2.bias"), aux=x2.view(M, H), Cp=x3p)
        self._last_out_planes = x3p
        saved = dict(planes=True, h2=h2, W=W, xh1=xh1, rs1=rs1, h1p=h1p, qkv=qkv, P=S, ctx=ctx, lse=lse, ctxp=ctxp, xh2=xh2, rs2=rs2,
                     h2p=h2p, gq=gq, fp=fp) if keep else None
        return x3, saved

    _dx_stat = None

    def _block_bwd_planes(self, i, dx3, s, defer, dx3_stat=None):
        """Backward of a block whose forward ran on planes: every GEMM operand is a plane set (the saved activations, the
        gradients split once where they are produced in fp32), all four weight + bias gradients in one grouped launch."""
        vm = "item_encoder.vision_model."
        name = lambda t: f"{vm}encoder.layers.{i}.{t}"
        P = lambda t, **kw: self.view(name(t), **kw)
        G = lambda t, **kw: self.view(name(t), grad=True, **kw)
        H, heads, d, T, _ = self._shape()
        n = dx3.shape[0]
        M = n * T
        ld, Tp, bh = 3 * H, (T + 3) & ~3, n * heads
        Wqkv, Wo, W1, W2 = s["W"]
        lead = (n, T)
        # fp16 two-plane operands (s["h2"]): each gradient is split with a scale found on the device; the one that leaves a GEMM
        # epilogue as planes (du) takes its scale from the bound |dx3 W2| * max |quick_gelu'| (= 1.0998)
        h2m = bool(s.get("h2"))
        split = (lambda t: ops.split_h2_auto([t])[0]) if h2m else ops.split_planes
        # round 6: the residual adds of the pre-LN block ride in the LayerNorm backward launches (dx2 = dx3 + LN2-backward, dx = dx2 +
        # LN1-backward: pxr_ln_bwd_res_f32, the same bits), and on fp16 two-plane operands those launches also leave the partial maxima
        # of the sums, so their splits need no statistics pass (ops.split_h2_parts) -- two launches less per site.  dx3_stat: the
        # maxima of dx3 when it came out of the next block's last launch.  PXR_TOWER_LN_RES=0: the separate launches (A/B)
        fuse_res = os.environ.get("PXR_TOWER_LN_RES", "1") != "0"
        n_parts = ops.ln_bwd_stat_parts(M) if (h2m and fuse_res) else 0
        stat = (lambda: torch.empty(n_parts, dtype=torch.float32, device=dx3.device)) if n_parts else (lambda: None)
        if n_parts and dx3_stat is not None:
            dx3p = ops.split_h2_parts(dx3.view(M, H), dx3_stat, n_parts)
        else:
            dx3p = split(dx3.view(M, H))
        pend = [(dx3p, s["fp"], G("mlp.fc2.weight"), G("mlp.fc2.bias"))]
        _, dup = ops.linear_bwd_input_planes(dx3p, W2, mul=s["gq"], want_fp32=False, want_planes=True, mul_bound=1.1)   # x quick_gelu'
        pend.append((dup, s["h2p"], G("mlp.fc1.weight"), G("mlp.fc1.bias")))
        dh2, _ = ops.linear_bwd_input_planes(dup, W1, lead_shape=lead)
        if fuse_res:
            st2 = stat()
            dx2, _ = ops.ln_bwd(0, dh2, s["xh2"], s["rs2"], P("layer_norm2.weight"), G("layer_norm2.weight"),
                                G("layer_norm2.bias"), defer=defer, res=dx3, stat=st2)
            dx2p = ops.split_h2_parts(dx2.view(M, H), st2, n_parts) if n_parts else split(dx2.view(M, H))
        else:
            dz2, _ = ops.ln_bwd(0, dh2, s["xh2"], s["rs2"], P("layer_norm2.weight"), G("layer_norm2.weight"),
                                G("layer_norm2.bias"), defer=defer)
            dx2 = ops.add(dx3, dz2)
            dx2p = split(dx2.view(M, H))
        pend.append((dx2p, s["ctxp"], G("self_attn.out_proj.weight"), G("self_attn.out_proj.bias")))
        dctx, _ = ops.linear_bwd_input_planes(dx2p, Wo, lead_shape=lead)
        if s["P"] is None:          # the forward ran the fused attention: so does the backward
            dqkv = ops.tower_attn_bwd(s["qkv"].view(M, ld), dctx.view(M, H), s["ctx"].view(M, H), s["lse"], n, T, heads, d,
                                      2 * H, 0, H, d ** -0.5).view(n, T, ld)
        else:
            dqkv = self._attn_bwd(s["qkv"], s["P"], dctx, n, T, heads, d)
        dqkvp = split(dqkv.view(M, ld))
        pend.append((dqkvp, s["h1p"], G("self_attn.k_proj.weight", span=3), G("self_attn.k_proj.bias", span=3)))
        dh1, _ = ops.linear_bwd_input_planes(dqkvp, Wqkv, lead_shape=lead)
        st1 = None
        if fuse_res:
            st1 = stat()
            dx, _ = ops.ln_bwd(0, dh1, s["xh1"], s["rs1"], P("layer_norm1.weight"), G("layer_norm1.weight"),
                               G("layer_norm1.bias"), defer=defer, res=dx2, stat=st1)
        else:
            dz1, _ = ops.ln_bwd(0, dh1, s["xh1"], s["rs1"], P("layer_norm1.weight"), G("layer_norm1.weight"),
                                G("layer_norm1.bias"), defer=defer)
            dx = ops.add(dx2, dz1)
        ops.grouped_dw_planes(pend)
        self._dx_stat = st1        # partial maxima of the returned gradient (None: not gathered): the previous block's first split
        return dx

    def _attn_bwd(self, qkv, Pm, dctx, n, T, heads, d):
        """d(q | k | v) of the materialized attention: four batched contractions + the row softmax backward."""
        H = heads * d
        ld, Tp, bh = 3 * H, (T + 3) & ~3, n * heads
        dqkv = torch.empty_like(qkv)
        dP = torch.empty_like(Pm)
        sP, sQ, sC = (heads * T * Tp, T * Tp), (T * ld, d), (T * H, d)
        ops.gemm_batched(False, False, T, d, T, Pm, 0, Tp, dctx, 0, H, dqkv, H, ld, bh, heads, sP, sC, sQ)    # dV = P^T dO
        ops.gemm_batched(True, True, T, T, d, dctx, 0, H, qkv, H, ld, dP, 0, Tp, bh, heads, sC, sQ, sP)      # dP = dO V^T
        ops.softmax_rows_bwd(Pm, dP, bh * T, T, Tp, d ** -0.5)                                            # dP := dS
        ops.gemm_batched(True, False, T, d, T, dP, 0, Tp, qkv, 0, ld, dqkv, 2 * H, ld, bh, heads, sP, sQ, sQ)  # dQ = dS K
        ops.gemm_batched(False, False, T, d, T, dP, 0, Tp, qkv, 2 * H, ld, dqkv, 0, ld, bh, heads, sP, sQ, sQ)  # dK = dS^T Q
        return dqkv

    def _block_fwd(self, i, x, keep, out_planes=False):
        self._last_out_planes = None
        if self._planes_on() and (not keep or os.environ.get("PXR_TOWER_TRAIN_PLANES", "1") != "0"):
            return self._block_fwd_planes(i, x, keep, out_planes)
        vm = "item_encoder.vision_model."
        P = lambda s, **kw: self.view(f"{vm}encoder.layers.{i}.{s}", **kw)
        H, heads, d, T, _ = self._shape()
        n = x.shape[0]
        h1, xh1, rs1 = ops.ln_residual_fwd(x, None, P("layer_norm1.weight"), P("layer_norm1.bias"), 1e-5, save=keep)
        qkv = ops.linear_fwd(h1, P("self_attn.k_proj.weight", span=3), P("self_attn.k_proj.bias", span=3))
        ctx, S = self._attn_fwd(qkv, n, T, heads, d, keep)
        x2 = ops.linear_epi(ctx, P("self_attn.out_proj.weight"), P("self_attn.out_proj.bias"), ops.EPI_BIAS_ADD, aux=x)
        h2, xh2, rs2 = ops.ln_residual_fwd(x2, None, P("layer_norm2.weight"), P("layer_norm2.bias"), 1e-5, save=keep)
        f, gq = ops.linear_epi(h2, P("mlp.fc1.weight"), P("mlp.fc1.bias"), ops.EPI_BIAS_QGELU_GRAD)
        x3 = ops.linear_epi(f, P("mlp.fc2.weight"), P("mlp.fc2.bias"), ops.EPI_BIAS_ADD, aux=x2)
        saved = dict(xh1=xh1, rs1=rs1, h1=h1, qkv=qkv, P=S, ctx=ctx, xh2=xh2, rs2=rs2, h2=h2, gq=gq, f=f) if keep else None
        return x3, saved

    def forward(self, images, need_grad: bool):
        """images fp32 [n, 3, Hi, Wi] -> (item vectors [n, D], saved state | None)."""
        self.ensure_packed()
        e = self.enc
        vm = "item_encoder.vision_model."
        H, heads, d, T, p = self._shape()
        n, c, Hi, Wi = images.shape
        if (Hi // p) * (Wi // p) != T - 1:
            raise ValueError(f"image size {Hi}x{Wi} does not give the encoder's {T - 1} patches of {p}x{p}")
        first = self.first_trainable_block() if need_grad else 10 ** 9
        # patch projection: conv(k = s = p, no bias) == im2col (a layout change) + GEMM on the fp32 MFMA
        patches = images.view(n, c, Hi // p, p, Wi // p, p).permute(0, 2, 4, 1, 3, 5).reshape(-1, c * p * p).contiguous()
        wp = self.view(vm + "embeddings.patch_embedding.weight").view(H, -1)
        pe = ops.linear_fwd(patches, wp, None).view(n, T - 1, H)
        x0 = ops.vit_embed(pe, self.view(vm + "embeddings.class_embedding"), self.view(vm + "embeddings.position_embedding.weight"))
        keep0 = first < 0
        x, xh0, rs0 = ops.ln_residual_fwd(x0, None, self.view(vm + "pre_layrnorm.weight"), self.view(vm + "pre_layrnorm.bias"),
                                          1e-5, save=keep0)
        n_layers = len(e.item_encoder.vision_model.encoder.layers)
        blocks = []
        head = e.head_layers()
        for _, _, a_ in head:
            if a_ is not None and not isinstance(a_, torch.nn.ReLU):
                raise NotImplementedError("native image encoder: rec_fc activation must be relu or none (overall/ViT.yaml: relu)")
        act_relu = isinstance(head[-1][2], torch.nn.ReLU)
        # opt-in (PXR_TOWER_H2_HEAD=1): the head's Linear over every token (method 'mean') on fp16 two-plane operands too -- the last
        # block writes its output as planes, the weight is split with a device-chosen scale.  Measured on MI355X (ViT-B/16, 69 344
        # tokens): 65.5 -> 66.2 ms per step -- the statistics pass over the head's gradient costs more than three products save on a
        # 768 x 512 weight; off by default.
        head_h2 = (e.native_method == "mean" and act_relu and len(head) == 1 and os.environ.get("PXR_TOWER_H2_HEAD", "0") == "1"
                   and self.view(head[0][0]).shape[0] % 32 == 0)
        for i in range(n_layers):
            x, s = self._block_fwd(i, x, keep=need_grad and i >= first, out_planes=head_h2 and i == n_layers - 1)
            blocks.append(s)
        xp_last = getattr(self, "_last_out_planes", None) if n_layers else None
        # head: rec_fc (Linear + activation) on every token then the token mean, or on the class token only
        W, b = self.view(head[-1][0]), self.view(head[-1][1])
        head_in = x if e.native_method == "mean" else x[:, 0, :].contiguous()
        xh_p = rs_p = None
        if e.native_method == "pool":            # pooler_output: post_layernorm of the class token (HF CLIPVisionTransformer)
            y_p, xh_p, rs_p = ops.ln_residual_fwd(head_in.view(n, 1, -1), None, self.view(vm + "post_layernorm.weight"),
                                                  self.view(vm + "post_layernorm.bias"), 1e-5, save=need_grad)
            head_in = y_p.view(n, -1)
        # an MLP head (fine_tune_arg.dnn_layers): the Linear + activation layers in front of the last one (same token set; the token
        # mean of 'mean' comes after the LAST activation, layers.py:125-128)
        hidden = []
        for wn, bn, a_ in head[:-1]:
            Wi, bi = self.view(wn), self.view(bn)
            a_i = ops.linear_epi(head_in, Wi, bi, ops.EPI_BIAS_RELU) if a_ is not None else ops.linear_fwd(head_in, Wi, bi)
            hidden.append((head_in, a_i, a_ is not None))
            head_in = a_i
        Wp = None
        if xp_last is not None:
            Wp, = ops.split_h2_auto([W], col_stats=True)
            act = torch.empty(n, T, W.shape[0], dtype=torch.float32, device=x.device)
            ops.gemm_planes(xp_last, Wp, act.view(n * T, -1), ops.EPI_BIAS_RELU, bias=b)
        else:
            act = ops.linear_epi(head_in, W, b, ops.EPI_BIAS_RELU) if act_relu else ops.linear_fwd(head_in, W, b)
        out = ops.token_mean(act) if e.native_method == "mean" else act
        saved = None
        if need_grad:
            saved = dict(n=n, first=first, blocks=blocks, x_last=x, head_in=head_in, act=act, act_relu=act_relu, xh_p=xh_p, rs_p=rs_p,
                         head=head, hidden=hidden,
                         patches=patches if keep0 else None, xh0=xh0, rs0=rs0, head_planes=(xp_last, Wp) if Wp is not None else None)
        return out, saved

    # ------------------------------------------------------------------------------------------ backward
    def _block_bwd(self, i, dx3, s, defer, dx3_stat=None):
        if s.get("planes"):
            return self._block_bwd_planes(i, dx3, s, defer, dx3_stat)
        vm = "item_encoder.vision_model."
        name = lambda t: f"{vm}encoder.layers.{i}.{t}"
        P = lambda t, **kw: self.view(name(t), **kw)
        G = lambda t, **kw: self.view(name(t), grad=True, **kw)
        H, heads, d, T, _ = self._shape()
        n = dx3.shape[0]
        M = n * T
        ld, Tp, bh = 3 * H, (T + 3) & ~3, n * heads
        pend = []
        # fc2 + residual:  x3 = x2 + f W2^T + b2
        pend.append((dx3.view(M, H), s["f"].view(M, -1), G("mlp.fc2.weight"), G("mlp.fc2.bias")))
        du = ops.linear_bwd_input(dx3, P("mlp.fc2.weight"), mul=s["gq"])               # x quick_gelu'(pre-activation)
        pend.append((du.view(M, -1), s["h2"].view(M, H), G("mlp.fc1.weight"), G("mlp.fc1.bias")))
        dh2 = ops.linear_bwd_input(du, P("mlp.fc1.weight"))
        dz2, _ = ops.ln_bwd(0, dh2, s["xh2"], s["rs2"], P("layer_norm2.weight"), G("layer_norm2.weight"),
                            G("layer_norm2.bias"), defer=defer)
        dx2 = ops.add(dx3, dz2)
        # attention + residual:  x2 = x + ctx Wo^T + bo
        pend.append((dx2.view(M, H), s["ctx"].view(M, H), G("self_attn.out_proj.weight"), G("self_attn.out_proj.bias")))
        dctx = ops.linear_bwd_input(dx2, P("self_attn.out_proj.weight"))
        dqkv = self._attn_bwd(s["qkv"], s["P"], dctx, n, T, heads, d)
        pend.append((dqkv.view(M, ld), s["h1"].view(M, H), G("self_attn.k_proj.weight", span=3),
                     G("self_attn.k_proj.bias", span=3)))
        dh1 = ops.linear_bwd_input(dqkv, P("self_attn.k_proj.weight", span=3))
        dz1, _ = ops.ln_bwd(0, dh1, s["xh1"], s["rs1"], P("layer_norm1.weight"), G("layer_norm1.weight"),
                            G("layer_norm1.bias"), defer=defer)
        dx = ops.add(dx2, dz1)
        ops.grouped_linear_bwd_weight(pend)      # all four weight + bias gradients of the block: one launch
        return dx

    def backward(self, d_out, saved):
        """d_out [n, D] -> fills the flat gradient buffer of the trainable parameters."""
        e = self.enc
        vm = "item_encoder.vision_model."
        H, heads, d, T, p = self._shape()
        n, first = saved["n"], saved["first"]
        G = lambda t: self.view(t, grad=True)
        defer = ops.DeferredReductions()
        d_out = d_out.contiguous()
        act, head_in = saved["act"], saved["head_in"]
        if e.native_method == "mean":
            if saved["act_relu"]:
                dact = ops.token_mean_relu_bwd(d_out, act)
            else:
                dact = ops.token_mean_relu_bwd(d_out, torch.ones_like(act))
            M = n * T
        else:
            dact = torch.where(act > 0, d_out, torch.zeros_like(d_out)) if saved["act_relu"] else d_out
            M = n
        D = dact.shape[-1]
        hp = saved.get("head_planes")
        head = saved["head"]
        wn_last, bn_last = head[-1][0], head[-1][1]
        dactp = None
        if hp is not None:       # the head ran on h2 operands: so do its weight and input gradients
            dactp, = ops.split_h2_auto([dact.view(M, D)])
            ops.grouped_dw_planes([(dactp, hp[0], G(wn_last), G(bn_last))])
        else:
            ops.grouped_linear_bwd_weight([(dact.view(M, D), head_in.reshape(M, -1), G(wn_last), G(bn_last))])
        n_layers = len(e.item_encoder.vision_model.encoder.layers)
        pool_ln = e.native_method == "pool" and saved["xh_p"] is not None and \
            e.item_encoder.vision_model.post_layernorm.weight.requires_grad
        only_head = first >= n_layers and not pool_ln     # nothing but rec_fc trains
        if only_head and not saved["hidden"]:
            defer.flush()
            return
        if dactp is not None:
            dxl, _ = ops.linear_bwd_input_planes(dactp, hp[1], lead_shape=(n, T))
        else:
            dxl = ops.linear_bwd_input(dact, self.view(wn_last))
        # back through the hidden layers of an MLP head, last to first (their weight gradients are needed whatever else trains)
        for k, ((h_in, a_i, relu_i), (wn, bn, _)) in enumerate(zip(reversed(saved["hidden"]), reversed(head[:-1]))):
            da = torch.where(a_i > 0, dxl.view_as(a_i), torch.zeros_like(a_i)) if relu_i else dxl.view_as(a_i)
            ops.grouped_linear_bwd_weight([(da.reshape(M, -1), h_in.reshape(M, -1), G(wn), G(bn))])
            if only_head and k == len(saved["hidden"]) - 1:
                break                             # the first layer's input gradient has no reader
            dxl = ops.linear_bwd_input(da, self.view(wn))
        if only_head:
            defer.flush()
            return
        if e.native_method == "pool":
            dxl, _ = ops.ln_bwd(0, dxl.view(n, 1, H), saved["xh_p"], saved["rs_p"], self.view(vm + "post_layernorm.weight"),
                                G(vm + "post_layernorm.weight"), G(vm + "post_layernorm.bias"), defer=defer)
            dxl = dxl.view(n, H)
            if first >= n_layers:                # every block frozen: the pooled head's LayerNorm (load.py:119-120 keeps it a
                defer.flush()                    # trainable parameter under 'pool') and rec_fc are all that train
                return
        if e.native_method == "mean":
            dx = dxl.view(n, T, H)
        else:
            dx = torch.zeros(n, T, H, dtype=torch.float32, device=dxl.device)
            dx[:, 0, :] = dxl
        st = None                                # partial maxima of dx when the launch that produced it gathered them
        for i in reversed(range(max(first, 0), n_layers)):
            self._dx_stat = None
            dx = self._block_bwd(i, dx, saved["blocks"][i], defer, dx3_stat=st)
            st = self._dx_stat
        if first < 0:                            # the embeddings train too (tune_scale < 5)
            dx0, _ = ops.ln_bwd(0, dx, saved["xh0"], saved["rs0"], self.view(vm + "pre_layrnorm.weight"),
                                G(vm + "pre_layrnorm.weight"), G(vm + "pre_layrnorm.bias"), defer=defer)
            gpos = G(vm + "embeddings.position_embedding.weight")
            ops.colsum(dx0.view(n, T * H), out=gpos.view(-1), defer=defer)
            defer.flush()
            G(vm + "embeddings.class_embedding").copy_(gpos[0])          # d cls = d pos[0] (both add into token 0)
            dpe = dx0[:, 1:, :].contiguous().view(n * (T - 1), H)
            gw = G(vm + "embeddings.patch_embedding.weight").view(H, -1)
            ops.linear_bwd_weight(dpe, saved["patches"], out=gw)
        else:
            defer.flush()


class _TowerFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, images, anchor, tower):
        out, saved = tower.forward(images, need_grad=True)
        ctx.tower, ctx.saved = tower, saved
        return out

    @staticmethod
    def backward(ctx, d_out):
        ctx.tower.backward(d_out, ctx.saved)
        ctx.saved = None
        return None, None, None


def run(enc, images):
    """MeanItemEncoder / ClsItemEncoder forward on a HIP tensor."""
    tower = enc._native
    tower.ensure_packed()
    trainable = any(p.requires_grad for p in enc.parameters())
    if torch.is_grad_enabled() and trainable:
        if enc._anchor is None or enc._anchor.device != images.device:
            enc._anchor = torch.zeros((), dtype=torch.float32, device=images.device, requires_grad=True)
        return _TowerFn.apply(images.contiguous(), enc._anchor, tower)
    out, _ = tower.forward(images.contiguous(), need_grad=False)
    return out
