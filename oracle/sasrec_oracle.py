"""CPU ORACLE -- TEST INFRASTRUCTURE ONLY.  Never imported by pixelrec_amd/ (the product path has no CPU
fallback); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

A plain-PyTorch fp32 restatement of the reference's SASRec (IDNet) hot path, written from the reference's
arithmetic (not its classes): every function cites the reference file:line it follows.  Parameters are a flat
dict keyed exactly like the reference `state_dict()`:

    item_embedding.weight [N,D]          position_embedding.weight [L,D]       LayerNorm.{weight,bias} [D]
    trm_encoder.layer.{i}.multi_head_attention.{query,key,value,dense}.{weight,bias}
    trm_encoder.layer.{i}.multi_head_attention.LayerNorm.{weight,bias}
    trm_encoder.layer.{i}.feed_forward.{dense_1,dense_2}.{weight,bias}
    trm_encoder.layer.{i}.feed_forward.LayerNorm.{weight,bias}

Pinning: the reference ships NO tests, so this oracle is pinned against golden vectors produced
by importing the reference itself in the dev container (oracle/make_golden.py -> tests/golden/*.npz); see
tests/test_oracle_golden.py.  Gradients come from torch autograd over this restatement (the reference's
backward IS autograd, trainer.py:122), the optimizer restates torch.optim.AdamW's single-tensor update.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import numpy as np
import torch

Params = Dict[str, torch.Tensor]


# ------------------------------------------------------------------------------------------------ init
def param_shapes(n_items: int, D: int, L: int, n_layers: int, inner: int):
    """Name -> shape in the reference's registration order (sasrec.py:31-45, layers.py:556-579,628-641)."""
    shapes = {"item_embedding.weight": (n_items, D), "position_embedding.weight": (L, D)}
    for i in range(n_layers):
        p = f"trm_encoder.layer.{i}."
        for nm in ("query", "key", "value", "dense"):
            shapes[p + f"multi_head_attention.{nm}.weight"] = (D, D)
            shapes[p + f"multi_head_attention.{nm}.bias"] = (D,)
        shapes[p + "multi_head_attention.LayerNorm.weight"] = (D,)
        shapes[p + "multi_head_attention.LayerNorm.bias"] = (D,)
        shapes[p + "feed_forward.dense_1.weight"] = (inner * D, D)
        shapes[p + "feed_forward.dense_1.bias"] = (inner * D,)
        shapes[p + "feed_forward.dense_2.weight"] = (D, inner * D)
        shapes[p + "feed_forward.dense_2.bias"] = (D,)
        shapes[p + "feed_forward.LayerNorm.weight"] = (D,)
        shapes[p + "feed_forward.LayerNorm.bias"] = (D,)
    shapes["LayerNorm.weight"] = (D,)
    shapes["LayerNorm.bias"] = (D,)
    return shapes


def synth_params(n_items: int, D: int, L: int, n_layers: int, inner: int, seed: int, std: float = 0.02,
                 perturb: bool = True) -> Params:
    """Deterministic parameters from numpy's PCG64 (stable across numpy versions), so that fixtures need not
    store weights.  Distribution follows `_init_weights` (sasrec.py:51-61): N(0, std) for every Linear /
    Embedding weight INCLUDING row 0 of the table; with perturb=True biases and LayerNorm affine are moved off
    their (0, 1) init so that parity tests exercise them."""
    rng = np.random.default_rng(seed)
    out = {}
    for name, shape in param_shapes(n_items, D, L, n_layers, inner).items():
        if name.endswith("LayerNorm.weight"):
            v = 1.0 + (0.1 * rng.standard_normal(shape) if perturb else 0.0)
        elif name.endswith(".bias"):
            v = (0.05 * rng.standard_normal(shape)) if perturb else np.zeros(shape)
        else:
            v = std * rng.standard_normal(shape)
        out[name] = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(v, shape)).astype(np.float32))
    return out


# ------------------------------------------------------------------------------------------------ pieces
def layer_norm(x, w, b, eps):
    """nn.LayerNorm over the last dim, biased variance (sasrec.py:45,81; layers.py:578,615,640,671)."""
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * w + b


def gelu_erf(x):
    """layers.py:651-660."""
    return x * 0.5 * (1.0 + torch.erf(x / math.sqrt(2.0)))


def attention_mask(item_seq_or_mask):
    """sasrec.py:119-126: 0 where key j <= query i and key is real, else -1e9.  [B,1,L,L] fp32."""
    m = item_seq_or_mask != 0
    Lq = m.shape[-1]
    ext = m[:, None, None, :].expand(-1, -1, Lq, -1)
    ext = torch.tril(ext)
    return torch.where(ext, 0.0, -1e9)


def _dropout(x, keep_mask, p):
    """nn.Dropout in training mode with an EXPLICIT keep mask (the ATen Philox stream cannot be reproduced;
    tests inject the HIP kernels' counter-hash mask, oracle/dropout_rng.py)."""
    if keep_mask is None or p == 0.0:
        return x
    return x * keep_mask.to(x.dtype) / (1.0 - p)


def hidden_act_fn(name):
    """FeedForward.get_hidden_act (layers.py:641-649; swish :662-663)."""
    return {"gelu": gelu_erf, "relu": torch.relu, "swish": lambda x: x * torch.sigmoid(x), "tanh": torch.tanh,
            "sigmoid": torch.sigmoid}[name]


def encoder_layer(p: Params, i: int, h, mask, n_heads, eps, drop=None, p_attn=0.0, p_hidden=0.0, trace=None, act="gelu"):
    """One TransformerLayer = MultiHeadAttention (layers.py:585-617) + FeedForward (layers.py:665-673)."""
    pre = f"trm_encoder.layer.{i}."
    B, L, D = h.shape
    d = D // n_heads
    lin = lambda x, nm: x @ p[pre + nm + ".weight"].t() + p[pre + nm + ".bias"]
    q = lin(h, "multi_head_attention.query").view(B, L, n_heads, d).permute(0, 2, 1, 3)
    k = lin(h, "multi_head_attention.key").view(B, L, n_heads, d).permute(0, 2, 3, 1)
    v = lin(h, "multi_head_attention.value").view(B, L, n_heads, d).permute(0, 2, 1, 3)
    s = torch.matmul(q, k) / math.sqrt(d) + mask                       # layers.py:595-601
    prob = torch.softmax(s, dim=-1)                                   # :604
    prob_d = _dropout(prob, None if drop is None else drop.get((i, "attn")), p_attn)   # :608
    ctx = torch.matmul(prob_d, v).permute(0, 2, 1, 3).contiguous().view(B, L, D)       # :609-612
    a = lin(ctx, "multi_head_attention.dense")                        # :613
    a = _dropout(a, None if drop is None else drop.get((i, "attn_out")), p_hidden)     # :614
    h1 = layer_norm(a + h, p[pre + "multi_head_attention.LayerNorm.weight"],
                    p[pre + "multi_head_attention.LayerNorm.bias"], eps)               # :615
    u = lin(h1, "feed_forward.dense_1")                               # :666
    f = hidden_act_fn(act)(u)                                         # :667
    f2 = lin(f, "feed_forward.dense_2")                               # :669
    f2 = _dropout(f2, None if drop is None else drop.get((i, "ffn_out")), p_hidden)    # :670
    h2 = layer_norm(f2 + h1, p[pre + "feed_forward.LayerNorm.weight"], p[pre + "feed_forward.LayerNorm.bias"], eps)
    if trace is not None:
        trace[f"layer{i}.probs"] = prob
        trace[f"layer{i}.ctx"] = ctx
        trace[f"layer{i}.attn_out"] = h1
        trace[f"layer{i}.ffn_out"] = h2
    return h2


def encode(p: Params, seq_ids, key_mask_src, cfg, drop=None, trace=None):
    """item ids [B,L] -> last-layer states [B,L,D] (sasrec.py:68-86 / :97-109)."""
    L = seq_ids.shape[1]
    x = p["item_embedding.weight"][seq_ids] + p["position_embedding.weight"][:L][None]   # :68,77-80 / :101-102
    h = layer_norm(x, p["LayerNorm.weight"], p["LayerNorm.bias"], cfg["layer_norm_eps"])   # :81
    h = _dropout(h, None if drop is None else drop.get("input"), cfg.get("hidden_dropout_prob", 0.0))  # :82
    if trace is not None:
        trace["input_emb"] = h
    mask = attention_mask(key_mask_src)                                                   # :84 / :106
    for i in range(cfg["n_layers"]):
        h = encoder_layer(p, i, h, mask, cfg["n_heads"], cfg["layer_norm_eps"], drop,
                          cfg.get("attn_dropout_prob", 0.0) if drop is not None else 0.0,
                          cfg.get("hidden_dropout_prob", 0.0) if drop is not None else 0.0, trace,
                          act=cfg.get("hidden_act", "gelu"))
    return h


def forward_loss(p: Params, items, masked_index, cfg, drop=None, trace=None):
    """SASRec.forward (sasrec.py:65-92): BPR-style loss against ONE sampled negative per position.
    items int64 [B,2,L+1], masked_index int64 [B,L] -> 0-dim fp32."""
    pos_ids, neg_ids = items[:, 0], items[:, 1]
    out = encode(p, pos_ids[:, :-1], masked_index, cfg, drop, trace)
    E = p["item_embedding.weight"]
    pos_score = (out * E[pos_ids[:, 1:]]).sum(-1)                     # :88
    neg_score = (out * E[neg_ids[:, 1:]]).sum(-1)                     # :89
    loss = -(torch.log((pos_score - neg_score).sigmoid() + 1e-8) * masked_index).sum(-1)   # :91
    if trace is not None:
        trace["output_embs"], trace["pos_score"], trace["neg_score"] = out, pos_score, neg_score
    return loss.mean(-1)                                              # :92


def loss_and_grads(p: Params, items, masked_index, cfg, drop=None):
    """Autograd over the restatement (the reference's backward is autograd too, trainer.py:122).  The table
    gradient is dense with row 0 zeroed, as nn.Embedding(padding_idx=0) produces (sasrec.py:31)."""
    leaf = {k: v.detach().clone().requires_grad_(True) for k, v in p.items()}
    loss = forward_loss(leaf, items, masked_index, cfg, drop)
    loss.backward()
    grads = {k: (v.grad if v.grad is not None else torch.zeros_like(v)) for k, v in leaf.items()}
    grads["item_embedding.weight"][0].zero_()                         # padding_idx=0
    return loss.detach(), grads


@torch.no_grad()
def predict(p: Params, item_seq, item_feature, cfg, trace=None):
    """SASRec.predict (sasrec.py:94-113): last-position state x full catalog -> [B,N]."""
    out = encode(p, item_seq, item_seq, cfg, None, trace)
    seq_output = out[:, -1]                                           # :110
    if trace is not None:
        trace["seq_output"] = seq_output
    return torch.matmul(seq_output, item_feature.t())                 # :112


@torch.no_grad()
def full_sort_scores(scores, history_u=None, history_i=None):
    """Trainer._full_sort_batch_eval masking (trainer.py:333-336)."""
    scores = scores.clone()
    scores[:, 0] = -np.inf
    if history_u is not None:
        scores[(history_u, history_i)] = -np.inf
    return scores


@torch.no_grad()
def topk_hits(scores, positive_u, positive_i, k):
    """Collector.eval_batch_collect 'rec.topk' (collector.py:131-139): [B, k+1] int = hit flags + #positives."""
    _, topk_idx = torch.topk(scores, k, dim=-1)
    pos_matrix = torch.zeros_like(scores, dtype=torch.int)
    pos_matrix[positive_u, positive_i] = 1
    pos_len = pos_matrix.sum(dim=1, keepdim=True)
    pos_idx = torch.gather(pos_matrix, dim=1, index=topk_idx)
    return torch.cat((pos_idx, pos_len), dim=1), topk_idx


def recall_ndcg(rec_topk: np.ndarray, topk_list):
    """Recall (metrics.py:135-136) and NDCG (metrics.py:162-178) per-rank SUMS (base_metric.py:61-67):
    returns {'recall@k': sum over users, 'ndcg@k': ...}; the trainer divides by the number of users after
    the all_gather (trainer.py:360-364,403-406)."""
    rec_topk = np.asarray(rec_topk)
    K = rec_topk.shape[1] - 1
    pos_index = rec_topk[:, :K].astype(bool)
    pos_len = rec_topk[:, K]
    recall = np.cumsum(pos_index, axis=1) / pos_len.reshape(-1, 1)
    len_rank = np.full_like(pos_len, K)
    idcg_len = np.where(pos_len > len_rank, len_rank, pos_len)
    iranks = np.zeros_like(pos_index, dtype=np.float64)
    iranks[:, :] = np.arange(1, K + 1)
    idcg = np.cumsum(1.0 / np.log2(iranks + 1), axis=1)
    for row, idx in enumerate(idcg_len):
        idcg[row, idx:] = idcg[row, idx - 1]
    dcg = np.cumsum(np.where(pos_index, 1.0 / np.log2(iranks + 1), 0), axis=1)
    ndcg = dcg / idcg
    res = {}
    for k in topk_list:
        res[f"recall@{k}"] = recall.sum(axis=0)[k - 1]
    for k in topk_list:
        res[f"ndcg@{k}"] = ndcg.sum(axis=0)[k - 1]
    return res


# ------------------------------------------------------------------------------------------------ AdamW
def adamw_step(param, grad, exp_avg, exp_avg_sq, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, weight_decay=0.1):
    """One torch.optim.AdamW update (torch/optim/adamw.py single-tensor path; called at trainer.py:125 with
    torch defaults betas=(0.9,0.999), eps=1e-8, trainer.py:102).  In-place on fp32 tensors; `step` is 1-based."""
    param.mul_(1 - lr * weight_decay)
    exp_avg.lerp_(grad, 1 - beta1)
    exp_avg_sq.mul_(beta2).addcmul_(grad, grad, value=1 - beta2)
    bias_correction1 = 1 - beta1 ** step
    bias_correction2 = 1 - beta2 ** step
    step_size = lr / bias_correction1
    denom = (exp_avg_sq.sqrt() / math.sqrt(bias_correction2)).add_(eps)
    param.addcdiv_(exp_avg, denom, value=-step_size)


class OracleTrainer:
    """zero_grad -> forward -> backward -> AdamW.step over ALL parameters incl. the whole table (dense
    weight-decay semantics, trainer.py:116-125, overall/ID.yaml:20-23).  Used as the CPU baseline in bench.py."""

    def __init__(self, params: Params, cfg, lr=1e-4, weight_decay=0.1, group_of=None):
        """group_of (optional): parameter name -> (lr, weight_decay), the two-group optimizers of trainer.py:66-96 -- split by
        'visual_encoder' in the name (:86-96) or by the `decay_check_name` fragment (:73-91; names there carry DDP's 'module.'
        prefix, which the caller's function accounts for)."""
        self.p = {k: v.clone() for k, v in params.items()}
        self.m = {k: torch.zeros_like(v) for k, v in params.items()}
        self.v = {k: torch.zeros_like(v) for k, v in params.items()}
        self.cfg, self.lr, self.wd, self.t = cfg, lr, weight_decay, 0
        self.group_of = group_of

    def step(self, items, masked_index, drop=None):
        loss, grads = loss_and_grads(self.p, items, masked_index, self.cfg, drop)
        self.t += 1
        for k in self.p:
            lr, wd = self.group_of(k) if self.group_of is not None else (self.lr, self.wd)
            adamw_step(self.p[k], grads[k], self.m[k], self.v[k], self.t, lr, weight_decay=wd)
        return loss
