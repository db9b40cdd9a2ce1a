"""MOSASRec (PixelNet) -- drop-in for `REC.model.PixelNet.mosasrec.MOSASRec` (code/REC/model/PixelNet/
mosasrec.py:9-128): the SASRec block over item vectors produced END-TO-END by a visual encoder.

Contract kept: `input_type`; `__init__(config, dataload)`; `forward((images [B, 2(L+1), 3, H, W], masked_index)) ->
loss`; `predict(item_seq, item_feature)`; `compute_item(images) -> [b, D]`; parameter names (`visual_encoder.*`,
`position_embedding`, `LayerNorm`, `trm_encoder.*`) so that the 'visual_encoder' / rec parameter-group split of
trainer.py:74-98 and reference checkpoints work.

How it maps onto the kernels: the encoder output of the batch, viewed [B*(L+1)*2, D], plays the role of the "table";
row ids (b, t, pos|neg) are a fixed arange pattern, so the SAME fused gather+pos+LN input kernel and BPR head kernels
run unchanged, and the gradient w.r.t. the encoder output is one elementwise kernel (pxr_mosasrec_emb_grad_f32) --
no sort, no sparse rows.  Autograd then continues into the visual encoder (torch ops on the trainable tail).
"""
from __future__ import annotations

import torch

from .. import ops
from ..utils.enum_type import InputType
from .seqcore import SeqRecCore
from .visual import load_model


class _PixelStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, item_emb, model, idx, masked_index):
        ctx.model = model
        table = item_emb.detach().reshape(-1, item_emb.shape[-1]).contiguous()
        ctx.table = table
        return model._forward_core(table, idx, masked_index, train=True).view(())

    @staticmethod
    def backward(ctx, grad_out):
        m = ctx.model
        dx0, coef, s = m._backward_core(grad_out, ctx.table)
        d_emb = ops.mosasrec_emb_grad(dx0, s["out"], coef)         # [B, L+1, 2, D]
        return d_emb, None, None, None


class MOSASRec(SeqRecCore):
    input_type = InputType.SEQ

    def __init__(self, config, dataload):
        super().__init__()
        self.pretrain_weights = config["pretrain_path"]
        self.item_num = dataload.item_num
        self.embedding_size = config["embedding_size"]
        self.visual_encoder = load_model(config=config)
        self._build_core(config)
        # mosasrec.py:50-53: position table N(0, init_range); encoder block through _init_weights; LayerNorm (1, 0)
        self.position_embedding.weight.data.normal_(mean=0.0, std=self.initializer_range)
        self.trm_encoder.apply(self._init_weights)
        self.LayerNorm.bias.data.zero_()
        self.LayerNorm.weight.data.fill_(1.0)
        if self.pretrain_weights:
            self.load_weights(self.pretrain_weights)
        self._idx_cache = {}

    def _row_ids(self, B, device):
        """ids of (b, t, pos|neg) in the flattened encoder output, laid out like SEQTrainDataset's `items` [B,2,L+1]."""
        key = (B, str(device))
        if key not in self._idx_cache:
            W = self.max_seq_length + 1
            base = (torch.arange(B, device=device).view(B, 1, 1) * W + torch.arange(W, device=device).view(1, 1, W)) * 2
            self._idx_cache[key] = (base + torch.arange(2, device=device).view(1, 2, 1)).contiguous()
        return self._idx_cache[key]

    def forward(self, interaction):
        """interaction = (images fp32 [B, 2(L+1), 3, H, W] ordered pos_0, neg_0, pos_1, neg_1, ... (trainset.py:145-165),
        masked_index int64 [B, L]) -> 0-dim loss (mosasrec.py:66-93)."""
        items, masked_index = interaction
        B = masked_index.shape[0]
        self._ensure_packed()
        item_emb = self.visual_encoder(items.flatten(0, 1)).view(B, -1, 2, self.embedding_size)   # mosasrec.py:69
        idx = self._row_ids(B, item_emb.device)
        masked_index = masked_index.contiguous()
        if torch.is_grad_enabled() and self.training:
            if not item_emb.requires_grad:   # fully frozen encoder: still drive the backward of the sequence block
                item_emb = item_emb + self._anchor * 0
            return _PixelStep.apply(item_emb, self, idx, masked_index)
        table = item_emb.detach().reshape(-1, self.embedding_size).contiguous()
        return self._forward_core(table, idx, masked_index, train=False).view(())

    @torch.no_grad()
    def encode_last(self, item_seq, item_feature):
        self._ensure_packed()
        item_seq = item_seq.contiguous()
        B, L = item_seq.shape
        feat = item_feature if item_feature.is_contiguous() else item_feature.contiguous()
        out, _ = self._encode(feat, item_seq, L, B, item_seq, L, train=False)      # mosasrec.py:102-109
        return out, out[:, -1]

    @torch.no_grad()
    def predict(self, item_seq, item_feature):
        """scores [B, N] from a precomputed item_feature table (mosasrec.py:95-114)."""
        out, last = self.encode_last(item_seq, item_feature)
        B, L, D = out.shape
        N = item_feature.shape[0]
        scores = torch.empty(B, N, dtype=torch.float32, device=out.device)
        ops.gemm(True, True, B, N, D, last, L * D, item_feature.contiguous(), D, scores, N, ops.EPI_NONE, use_ws=False)
        return scores

    @torch.no_grad()
    def compute_item(self, item):
        return self.visual_encoder(item)                                              # mosasrec.py:117-119
