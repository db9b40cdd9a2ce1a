"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's GRU4Rec (code/REC/model/IDNet/gru4rec.py:50-87), the GRU
written out step by step (torch.nn.GRU's documented equations with bias=False: r | z | n gate order).  Imported by tests/
only.  Pinned against the reference itself: tests/golden/gru4rec_tiny.npz is written by oracle/make_golden_gru4rec.py from
`REC.model.IDNet.gru4rec.GRU4Rec` run unmodified; tests/test_gru4rec_golden.py checks this file against it.
"""
import torch


def gru_stack(params, x, n_layers):
    """x [B, L, E] -> h of the top layer [B, L, H]  (gru4rec.py:60 / :77, h_0 = 0)."""
    B, L, _ = x.shape
    for k in range(n_layers):
        w_ih, w_hh = params[f"gru_layers.weight_ih_l{k}"], params[f"gru_layers.weight_hh_l{k}"]
        H = w_hh.shape[1]
        h = x.new_zeros(B, H)
        outs = []
        for t in range(L):
            gi, gh = x[:, t] @ w_ih.t(), h @ w_hh.t()
            r = torch.sigmoid(gi[:, :H] + gh[:, :H])
            z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
            h = (1 - z) * n + z * h
            outs.append(h)
        x = torch.stack(outs, dim=1)
    return x


def encode(params, seq_ids, n_layers, emb_keep=None, p_drop=0.0):
    """item ids [B, L] -> dense(GRU(emb_dropout(E[ids]))) [B, L, E]  (gru4rec.py:58-62).  emb_keep: boolean keep mask [B, L, E] of
    the embedding dropout (None = inactive: evaluation, or the shipped dropout_prob 0); kept values are scaled by 1 / (1 - p)."""
    x = params["item_embedding.weight"][seq_ids]
    if emb_keep is not None:
        x = torch.where(emb_keep, x / (1.0 - p_drop), torch.zeros_like(x))
    out = gru_stack(params, x, n_layers)
    return out @ params["dense.weight"].t() + params["dense.bias"]


def forward_loss(params, items, masked_index, n_layers, emb_keep=None, p_drop=0.0):
    """gru4rec.py:50-68."""
    pos_ids, neg_ids = items[:, 0], items[:, 1]
    out = encode(params, pos_ids[:, :-1], n_layers, emb_keep, p_drop)
    E = params["item_embedding.weight"]
    pos = (out * E[pos_ids[:, 1:]]).sum(-1)
    neg = (out * E[neg_ids[:, 1:]]).sum(-1)
    loss = -(torch.log((pos - neg).sigmoid() + 1e-8) * masked_index).sum(-1)
    return loss.mean(-1)


@torch.no_grad()
def predict(params, item_seq, item_feature, n_layers):
    """gru4rec.py:71-82."""
    out = encode({**params, "item_embedding.weight": item_feature}, item_seq, n_layers)
    return out[:, -1] @ item_feature.t()


def forward_loss_rows(params, item_emb, masked_index, n_layers):
    """PixelNet's MOGRU4Rec.forward after the image encoder (code/REC/model/PixelNet/mogru4rec.py:45-63): item_emb
    [B, L+1, 2, D] with pos | neg interleaved -- the same composition as MOSASRec (oracle/mosasrec_oracle.py)."""
    pos, neg = item_emb[:, :, 0], item_emb[:, :, 1]
    out = gru_stack(params, pos[:, :-1], n_layers) @ params["dense.weight"].t() + params["dense.bias"]
    ps, ns = (out * pos[:, 1:]).sum(-1), (out * neg[:, 1:]).sum(-1)
    return (-(torch.log((ps - ns).sigmoid() + 1e-8) * masked_index).sum(-1)).mean(-1)
