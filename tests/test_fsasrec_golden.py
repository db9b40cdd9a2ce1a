"""oracle/fsasrec_oracle.py pinned against the reference: tests/golden/fsasrec_tiny.npz holds what
`REC.model.ViNet.fsasrec.FSASRec` (run unmodified by oracle/make_golden_fsasrec.py) computes for its three item encoders
(freeze / hybrid / semantic, plus the MLPLayers projection): loss, every parameter gradient, compute_item_all, predict."""
import os

import numpy as np
import pytest
import torch

from oracle import fsasrec_oracle as FO

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "fsasrec_tiny.npz"))
N, F, D, L, H, NL, B, C = [int(x) for x in G["dims"]]
CFG = {"n_layers": NL, "n_heads": H, "layer_norm_eps": 1e-12}
KINDS = {"fix": "fix", "fixmlp": "fix", "hybrid": "hybrid", "semantic": "semantic"}


def golden_case(case, dtype=torch.float64):
    params = {k[len(case) + 7:]: torch.from_numpy(G[k]).to(dtype) for k in G.files if k.startswith(case + "/param/")}
    kind = KINDS[case]
    table = FO.shifted_codes(torch.from_numpy(G["codes"])) if kind == "semantic" else torch.from_numpy(G["feats"]).to(dtype)
    return kind, params, table


@pytest.mark.parametrize("case", list(KINDS))
def test_oracle_reproduces_the_reference(case):
    kind, params, table = golden_case(case)
    for p in params.values():
        p.requires_grad_(True)
    items, mask = torch.from_numpy(G["items"]), torch.from_numpy(G["masked_index"])
    loss = FO.forward_loss(kind, params, table, items, mask, CFG)
    assert abs(float(loss) - float(G[case + "/loss"])) < 2e-6
    loss.backward()
    n_checked = 0
    for name, p in params.items():
        want = torch.from_numpy(G[f"{case}/grad/{name}"]).double()
        got = p.grad if p.grad is not None else torch.zeros_like(p)
        if name == "item_embedding.pq_code_embedding.weight":
            got = got.clone(); got[0] = 0                      # padding_idx = 0: the reference's nn.Embedding drops that row's gradient
        assert (got - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item()), name
        n_checked += 1
    assert n_checked == len(params) and any(k.startswith("item_embedding.") for k in params)
    with torch.no_grad():
        feat = FO.compute_item_all(kind, params, table)
        assert (feat - torch.from_numpy(G[case + "/item_all"]).double()).abs().max().item() < 2e-6
        scores = FO.predict(kind, params, table, torch.from_numpy(G["item_seq"]), CFG)
        assert (scores - torch.from_numpy(G[case + "/scores"]).double()).abs().max().item() < 5e-6


def test_golden_covers_the_edge_cases():
    assert (G["items"][:, 0, :] == 0).any() and (G["masked_index"] == 0).any()        # left padding present
    assert G["codes"][:, 0].max() == 6 and (G["codes"][0] == 0).all()                 # code_cap from column 0; padding item
    assert any(k.endswith("rec_fc.mlp_layers.4.weight") for k in G.files)             # the dnn_layers (MLPLayers) projection
