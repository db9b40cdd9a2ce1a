"""oracle/gru4rec_oracle.py pinned against the reference: tests/golden/gru4rec_tiny.npz holds what
`REC.model.IDNet.gru4rec.GRU4Rec` (run unmodified by oracle/make_golden_gru4rec.py) computes: loss, every parameter gradient
(two GRU layers, left-padded sequences), predict scores."""
import os

import numpy as np
import torch

from oracle import gru4rec_oracle as GO

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "gru4rec_tiny.npz"))
N, E, MULT, NL, L, B = [int(x) for x in G["dims"]]


def golden_params(dtype=torch.float64):
    return {k[6:]: torch.from_numpy(G[k]).to(dtype) for k in G.files if k.startswith("param/")}


def test_oracle_reproduces_the_reference():
    params = golden_params()
    assert list(params) == ["item_embedding.weight", "gru_layers.weight_ih_l0", "gru_layers.weight_hh_l0",
                            "gru_layers.weight_ih_l1", "gru_layers.weight_hh_l1", "dense.weight", "dense.bias"]
    for p in params.values():
        p.requires_grad_(True)
    items, mask = torch.from_numpy(G["items"]), torch.from_numpy(G["masked_index"])
    loss = GO.forward_loss(params, items, mask, NL)
    assert abs(float(loss.detach()) - float(G["loss"])) < 2e-6
    loss.backward()
    for name, p in params.items():
        want = torch.from_numpy(G["grad/" + name]).double()
        got = p.grad.clone()
        if name == "item_embedding.weight":
            got[0] = 0                                             # padding_idx = 0 (gru4rec.py:25)
        assert (got - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item()), name
    with torch.no_grad():
        p = golden_params()
        scores = GO.predict(p, torch.from_numpy(G["item_seq"]), p["item_embedding.weight"], NL)
        assert (scores - torch.from_numpy(G["scores"]).double()).abs().max().item() < 5e-6
    assert (G["items"][:, 0, :] == 0).any() and (G["masked_index"] == 0).any()
