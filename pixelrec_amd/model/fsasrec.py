"""FSASRec (ViNet) -- drop-in for `REC.model.ViNet.fsasrec.FSASRec` (code/REC/model/ViNet/fsasrec.py:10-141; the
FreezeModel copy of the same class is its `freeze_model` branch): the SASRec block over item vectors that come from a
FROZEN per-item feature matrix (pre-extracted image features, `v_feat_path`) through a small trainable projection,
optionally concatenated with a trainable id embedding (`hybrid_model`), or from product-quantisation codes through a code
embedding (`semantic_model`) -- `load_weights`, code/REC/model/load.py:167-188, and the three encoders of
code/REC/model/layers.py:141-232.

Contract kept: `input_type`; `__init__(config, dataload)`; `forward((items int64 [B, 2, L+1], masked_index [B, L])) ->
loss`; `predict(item_seq, item_feature)`; `compute_item_all()`; parameter names (`item_embedding.rec_fc.*`,
`item_embedding.item_id_embedding.weight`, `item_embedding.pq_code_embedding.weight`, `position_embedding`, `LayerNorm`,
`trm_encoder.*`), so reference checkpoints and the modal / rec parameter-group split of trainer.py:74-98 (every FSASRec
parameter is a 'rec' parameter) work.

How it maps onto the kernels: the encoder output of the batch's 2 B (L+1) item ids plays the role of the "table" of the
shared sequence block (as in PixelNet's MOSASRec), so the fused gather + position + LayerNorm input kernel, attention,
GEMMs and the BPR head run unchanged and the gradient w.r.t. the item vectors is one elementwise kernel
(pxr_mosasrec_emb_grad_f32).  The encoders themselves are a row gather (pxr_embed_gather_f32) and the library's GEMMs with
the bias + ReLU epilogue; their weight gradients go through the grouped weight-gradient launch, the (dense, as in the
reference's torch AdamW) gradients of the id / code tables through the deterministic sort-and-segment row sums of
csrc/embed_grad.hip.  Every trainable parameter lives in the model's flat buffer, so PxrAdamW updates it with the rec
group's lr / weight decay -- exactly the reference's grouping.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from .. import ops
from ..utils.enum_type import InputType
from .seqcore import SeqRecCore


class MLPLayers(nn.Module):
    """Parameter container with the reference's module names (layers.py:239-295: [Dropout, Linear, activation] per layer;
    bn is never enabled by load_weights)."""

    def __init__(self, layers, dropout=0.0, activation="relu"):
        super().__init__()
        if dropout:
            raise NotImplementedError("MLPLayers dropout > 0 is not built (load_weights never sets it, load.py:167-188)")
        mods = []
        for i, o in zip(layers[:-1], layers[1:]):
            mods += [nn.Dropout(p=dropout), nn.Linear(i, o), nn.ReLU()]
        self.mlp_layers = nn.Sequential(*mods)


def _xavier_init(module):
    if isinstance(module, (nn.Linear, nn.Embedding)):
        nn.init.xavier_normal_(module.weight.data)
    if isinstance(module, nn.Linear) and module.bias is not None:
        nn.init.constant_(module.bias.data, 0)


def _rec_fc(input_dim, output_dim, dnn_layers):
    if dnn_layers:
        fc = MLPLayers([input_dim] + list(dnn_layers) + [output_dim], activation="relu")
    else:
        fc = nn.Sequential(nn.Linear(input_dim, output_dim), nn.ReLU())
    fc.apply(_xavier_init)
    return fc


class FIXItemEncoder(nn.Module):
    """layers.py:141-170: rec_fc(item_weights[x]); item_weights is a plain tensor (not in the state_dict), as there."""
    kind = "fix"

    def __init__(self, weight_path, device, output_dim, act_name="relu", dnn_layers=None):
        super().__init__()
        self.item_weights = torch.tensor(np.load(weight_path), dtype=torch.float32).to(device)
        self.rec_fc = _rec_fc(self.item_weights.shape[-1], output_dim, dnn_layers)


class HYItemEncoder(nn.Module):
    """layers.py:173-205: rec_fc(cat(item_weights[x], item_id_embedding(x)))."""
    kind = "hybrid"

    def __init__(self, weight_path, device, output_dim, item_num, act_name="relu", dnn_layers=None):
        super().__init__()
        self.item_weights = torch.tensor(np.load(weight_path), dtype=torch.float32).to(device)
        self.item_id_embedding = nn.Embedding(item_num, output_dim)
        self.rec_fc = _rec_fc(self.item_weights.shape[-1] + output_dim, output_dim, dnn_layers)
        nn.init.xavier_normal_(self.item_id_embedding.weight.data)


class SEMATICItemEncoder(nn.Module):
    """layers.py:209-232 (the reference's spelling): mean over the code positions of pq_code_embedding(pq_codes[x]); the
    codes of position j are shifted by j (1 + code_cap) into one table whose row 0 is the padding row."""
    kind = "semantic"

    def __init__(self, weight_path, device, output_dim, act_name="relu", dnn_layers=None):
        super().__init__()
        codes = torch.tensor(np.load(weight_path)).to(device).long()
        self.code_dim = codes.shape[-1]
        self.code_cap = int(codes[:, 0].max() + 1)
        acc = torch.tensor(np.cumsum([0] + [1 + self.code_cap] * (self.code_dim - 1))).to(device)
        self.pq_codes = codes + acc
        self.pq_code_embedding = nn.Embedding(self.code_dim * (1 + self.code_cap), output_dim, padding_idx=0)
        nn.init.xavier_normal_(self.pq_code_embedding.weight.data)


def load_weights(config):
    """load.py:167-188."""
    device, output_dim, dnn_layers = config["device"], config["embedding_size"], config["dnn_layers"]
    if config["semantic_model"]:
        return SEMATICItemEncoder(weight_path=config["semantic_id_path"], device=device, output_dim=output_dim,
                                  dnn_layers=dnn_layers)
    if config["hybrid_model"]:
        return HYItemEncoder(weight_path=config["v_feat_path"], device=device, output_dim=output_dim,
                             item_num=config["item_num"], dnn_layers=dnn_layers)
    if config["freeze_model"]:
        return FIXItemEncoder(weight_path=config["v_feat_path"], device=device, output_dim=output_dim, dnn_layers=dnn_layers)
    raise ValueError("FSASRec needs one of semantic_model / hybrid_model / freeze_model (load.py:167-188 returns nothing "
                     "otherwise and the reference then fails on its first forward)")


class _FeatStep(torch.autograd.Function):
    @staticmethod
    def forward(ctx, anchor, model, ids, idx, masked_index):
        table, saved = model._encode_items(ids, keep=True)
        ctx.model, ctx.table, ctx.saved = model, table, saved
        return model._forward_core(table, idx, masked_index, train=True).view(())

    @staticmethod
    def backward(ctx, grad_out):
        m = ctx.model
        dx0, coef, s = m._backward_core(grad_out, ctx.table)
        d_emb = ops.mosasrec_emb_grad(dx0, s["out"], coef)            # [B, L+1, 2, D] = the order the ids were encoded in
        m._encode_items_bwd(d_emb.view(-1, d_emb.shape[-1]), ctx.saved)
        ctx.saved = ctx.table = None
        return None, None, None, None, None


class FSASRec(SeqRecCore):
    input_type = InputType.SEQ

    def __init__(self, config, dataload):
        super().__init__()
        self.item_num = dataload.item_num
        self.config = config
        cfg = {k: (config[k] if k in config else None) for k in
               ("device", "embedding_size", "dnn_layers", "semantic_model", "hybrid_model", "freeze_model", "v_feat_path",
                "semantic_id_path")}
        cfg["item_num"] = dataload.item_num                            # fsasrec.py:33
        self.item_embedding = load_weights(cfg)
        self._build_core(config)
        self.apply(self._init_weights)                                 # fsasrec.py:54: re-draws the encoder's weights too
        self._idx_cache = {}

    # ------------------------------------------------------------------------------------------ flat packing
    def _enc_linears(self):
        fc = getattr(self.item_embedding, "rec_fc", None)
        if fc is None:
            return []
        seq = fc.mlp_layers if isinstance(fc, MLPLayers) else fc
        return [m for m in seq if isinstance(m, nn.Linear)]

    def _flat_specs(self):
        specs = super()._flat_specs()
        for k, lin in enumerate(self._enc_linears()):
            specs += [(f"enc.{k}.w", lin.weight), (f"enc.{k}.b", lin.bias)]
        enc = self.item_embedding
        if enc.kind == "hybrid":
            specs.append(("enc.id", enc.item_id_embedding.weight))
        if enc.kind == "semantic":
            specs.append(("enc.pq", enc.pq_code_embedding.weight))
        return specs

    def encoder_parameter_names(self):
        """{reference parameter name: flat-buffer key} of the item encoder, in the order the reference registers them
        (what torch.optim.AdamW numbers its state in; optim.reference_rec_parameter_names)."""
        enc, out = self.item_embedding, {}
        if enc.kind == "hybrid":
            out["item_embedding.item_id_embedding.weight"] = "enc.id"
        if enc.kind == "semantic":
            out["item_embedding.pq_code_embedding.weight"] = "enc.pq"
        lins = self._enc_linears()
        names = {id(p): n for n, p in self.named_parameters()}
        for k, lin in enumerate(lins):
            out[names[id(lin.weight)]] = f"enc.{k}.w"
            out[names[id(lin.bias)]] = f"enc.{k}.b"
        return out

    # ------------------------------------------------------------------------------------------ item encoder
    def _encode_items(self, ids, keep):
        """ids int64 [n] -> (item vectors fp32 [n, D], what the backward needs | None)."""
        enc = self.item_embedding
        ids = ids.contiguous()
        if enc.kind == "semantic":
            codes = enc.pq_codes[ids].contiguous()                                        # [n, C]
            rows = ops.embed_gather(self._p("enc.pq"), codes)                             # [n, C, D]
            return ops.token_mean(rows), ((codes,) if keep else None)
        x = ops.embed_gather(enc.item_weights, ids)                                       # [n, F]
        if enc.kind == "hybrid":
            x = torch.cat((x, ops.embed_gather(self._p("enc.id"), ids)), dim=-1)          # layers.py:202-204
        acts = [x]
        for k in range(len(self._enc_linears())):
            acts.append(ops.linear_epi(acts[-1], self._p(f"enc.{k}.w"), self._p(f"enc.{k}.b"), ops.EPI_BIAS_RELU))
        return acts[-1], ((ids, acts) if keep else None)

    def _encode_items_bwd(self, d_out, saved):
        """d_out [n, D] -> fills the flat gradient views of the encoder's parameters (overwrites, like every backward here)."""
        enc = self.item_embedding
        D = d_out.shape[-1]
        if enc.kind == "semantic":
            (codes,) = saved
            n, C = codes.shape
            rows = (d_out / C).unsqueeze(1).expand(n, C, D).reshape(n * C, D).contiguous()   # d mean: 1/C to every code row
            self._dense_rows_grad(self._p("enc.pq", grad=True), codes.reshape(-1), rows, drop_row0=True)   # padding_idx = 0
            return
        ids, acts = saved
        n_lin = len(acts) - 1
        d = d_out
        for k in reversed(range(n_lin)):
            dz = (d * (acts[k + 1] > 0)).contiguous()                                    # ReLU
            ops.grouped_linear_bwd_weight([(dz, acts[k], self._p(f"enc.{k}.w", grad=True), self._p(f"enc.{k}.b", grad=True))])
            if k > 0 or enc.kind == "hybrid":
                d = ops.linear_bwd_input(dz, self._p(f"enc.{k}.w"))
        if enc.kind == "hybrid":
            F = enc.item_weights.shape[-1]
            self._dense_rows_grad(self._p("enc.id", grad=True), ids, d[:, F:].contiguous(), drop_row0=False)

    def _dense_rows_grad(self, gtable, ids, rows, drop_row0):
        """gtable[N, D] = sum of `rows` per id (dense gradient of an embedding table, deterministic): the sort-and-segment
        row sums of the sparse path, scattered into the zeroed dense view.  drop_row0: row 0 is a padding_idx row (the code
        table; the reference's nn.Embedding gives it no gradient).  The id table of the hybrid encoder has no padding_idx
        (layers.py:177), so its row 0 keeps whatever gradient reaches it: ids are shifted by one around the row-sum kernel,
        which treats id 0 as "skip"."""
        shift = 0 if drop_row0 else 1
        sp = ops.embed_grad_rows((ids + shift).contiguous(), rows, gtable.shape[0] + shift)
        gtable.zero_()
        ar = torch.arange(sp.cap, device=ids.device, dtype=torch.int32)
        pos = torch.where(ar < sp.n, (sp.idx - shift).to(torch.int32), torch.full_like(ar, -1))
        ops.scatter_rows(sp.rows, pos, gtable, 0)

    # ------------------------------------------------------------------------------------------ model interface
    def _row_ids(self, B, device):
        """ids of (b, t, pos | neg) in the flattened encoder output [B, L+1, 2, D], shaped like `items` [B, 2, L+1]."""
        key = (B, str(device))
        if key not in self._idx_cache:
            W = self.max_seq_length + 1
            base = (torch.arange(B, device=device).view(B, 1, 1) * W + torch.arange(W, device=device).view(1, 1, W)) * 2
            self._idx_cache[key] = (base + torch.arange(2, device=device).view(1, 2, 1)).contiguous()
        return self._idx_cache[key]

    def forward(self, interaction):
        """interaction = (items int64 [B, 2, L+1], masked_index int64 [B, L]) -> 0-dim loss (fsasrec.py:66-92)."""
        items, masked_index = interaction
        if items.dim() != 3 or items.shape[1] != 2 or items.shape[2] != self.max_seq_length + 1:
            raise ValueError(f"items must be [B, 2, {self.max_seq_length + 1}], got {tuple(items.shape)}")
        self._ensure_packed()
        B = items.shape[0]
        ids = items.transpose(1, 2).reshape(-1)                 # (b, t, pos | neg): the row order of the block the core reads
        idx = self._row_ids(B, items.device)
        masked_index = masked_index.contiguous()
        if torch.is_grad_enabled() and self.training:
            return _FeatStep.apply(self._anchor, self, ids, idx, masked_index)
        table, _ = self._encode_items(ids, keep=False)
        return self._forward_core(table, idx, masked_index, train=False).view(())

    @torch.no_grad()
    def encode_last(self, item_seq, item_feature):
        self._ensure_packed()
        item_seq = item_seq.contiguous()
        B, L = item_seq.shape
        feat = item_feature if item_feature.is_contiguous() else item_feature.contiguous()
        out, _ = self._encode(feat, item_seq, L, B, item_seq, L, train=False)             # fsasrec.py:97-109
        return out, out[:, -1]

    @torch.no_grad()
    def predict(self, item_seq, item_feature):
        """scores [B, N] (fsasrec.py:94-113).  The reference re-encodes the sequence's items (`self.item_embedding(item_seq)`);
        item_feature = compute_item_all() holds the same vectors, so its rows are gathered instead."""
        out, last = self.encode_last(item_seq, item_feature)
        B, L, D = out.shape
        N = item_feature.shape[0]
        scores = torch.empty(B, N, dtype=torch.float32, device=out.device)
        ops.gemm(True, True, B, N, D, last, L * D, item_feature.contiguous(), D, scores, N, ops.EPI_NONE, use_ws=False)
        return scores

    @torch.no_grad()
    def compute_item_all(self):
        """[N, D] item vectors of the whole catalogue (fsasrec.py:115-122), encoded in chunks."""
        self._ensure_packed()
        enc = self.item_embedding
        N = enc.pq_codes.shape[0] if enc.kind == "semantic" else enc.item_weights.shape[0]
        dev = self._flat.device
        out = []
        for s in range(0, N, 65536):
            ids = torch.arange(s, min(N, s + 65536), device=dev, dtype=torch.int64)
            out.append(self._encode_items(ids, keep=False)[0])
        return torch.cat(out) if len(out) > 1 else out[0]
