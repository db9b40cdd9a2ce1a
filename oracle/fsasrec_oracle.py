"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's FSASRec item encoders and forward loss.  Imported by tests/
only (never by pixelrec_amd/).  Pinned against the reference itself: tests/golden/fsasrec_tiny.npz is written by
oracle/make_golden_fsasrec.py from `REC.model.ViNet.fsasrec.FSASRec` run unmodified, and tests/test_fsasrec_golden.py checks
this file against it.

Follows:  code/REC/model/layers.py:141-170 (FIXItemEncoder), :173-205 (HYItemEncoder), :209-232 (SEMATICItemEncoder),
          code/REC/model/ViNet/fsasrec.py:66-92 (forward), :94-113 (predict), :115-122 (compute_item_all).
"""
import torch

from oracle import mosasrec_oracle as MO
from oracle import sasrec_oracle as O


def _linears(params):
    """(weight, bias) of the encoder's Linear layers in order: rec_fc.0 or rec_fc.mlp_layers.{1,4,...} (layers.py:146-153)."""
    keys = sorted((k for k in params if k.startswith("item_embedding.rec_fc.") and k.endswith(".weight")),
                  key=lambda k: int([t for t in k.split(".") if t.isdigit()][-1]))
    return [(params[k], params[k[:-6] + "bias"]) for k in keys]


def item_vectors(kind, params, table, ids):
    """kind 'fix' | 'hybrid': table = item_weights [N, F]; 'semantic': table = the SHIFTED pq codes [N, C] (layers.py:216-219)."""
    if kind == "semantic":
        return params["item_embedding.pq_code_embedding.weight"][table[ids]].mean(dim=-2)      # layers.py:228-232
    x = table[ids]
    if kind == "hybrid":
        x = torch.cat((x, params["item_embedding.item_id_embedding.weight"][ids]), dim=-1)     # layers.py:201-205
    for w, b in _linears(params):
        x = torch.relu(x @ w.t() + b)                                                          # Linear + ReLU per layer
    return x


def shifted_codes(codes):
    """layers.py:212-219: code j of an item is moved into its own (1 + code_cap)-wide range."""
    codes = codes.long()
    cap = int(codes[:, 0].max() + 1)
    acc = torch.cumsum(torch.tensor([0] + [1 + cap] * (codes.shape[-1] - 1)), 0)
    return codes + acc


def forward_loss(kind, params, table, items, masked_index, cfg):
    """fsasrec.py:66-92: items int64 [B, 2, L+1] (row 0 positives, row 1 negatives)."""
    emb = item_vectors(kind, params, table, items)                       # [B, 2, L+1, D]
    return MO.forward_loss(params, emb.permute(0, 2, 1, 3), masked_index, cfg)


def compute_item_all(kind, params, table):
    return item_vectors(kind, params, table, torch.arange(table.shape[0]))


def predict(kind, params, table, item_seq, cfg):
    """fsasrec.py:94-113 (dropout inactive in eval)."""
    feat = compute_item_all(kind, params, table)
    # the sequence is re-encoded through the item encoder (fsasrec.py:101) -- the same vectors as the rows of `feat`
    return O.predict({**params, "item_embedding.weight": feat}, item_seq, feat, cfg)
