"""oracle/nextitnet_oracle.py pinned against the reference: tests/golden/nextitnet_tiny.npz holds what
`REC.model.IDNet.nextitnet.NextItNet` (run unmodified by oracle/make_golden_nextitnet.py) computes, with and without its
final_layer: loss, every parameter gradient (4 residual blocks, dilations 1,2,1,2 and their doubles, left-padded sequences),
predict scores."""
import os

import numpy as np
import pytest
import torch

from oracle import nextitnet_oracle as NO

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "nextitnet_tiny.npz"))
N, E, K, BLOCKS, L, B = [int(x) for x in G["dims"]]
DIL = [int(d) for d in G["dilations"]] * BLOCKS


def golden_params(case, dtype=torch.float64):
    pre = case + "/param/"
    return {k[len(pre):]: torch.from_numpy(G[k]).to(dtype) for k in G.files if k.startswith(pre)}


@pytest.mark.parametrize("case", ["plain", "final"])
def test_oracle_reproduces_the_reference(case):
    params = golden_params(case)
    assert ("final_layer.weight" in params) == (case == "final") and "residual_blocks.3.ln2.bias" in params
    for p in params.values():
        p.requires_grad_(True)
    items, mask = torch.from_numpy(G["items"]), torch.from_numpy(G["masked_index"])
    loss = NO.forward_loss(params, items, mask, DIL)
    assert abs(float(loss.detach()) - float(G[case + "/loss"])) < 2e-6
    loss.backward()
    for name, p in params.items():
        want = torch.from_numpy(G[f"{case}/grad/{name}"]).double()
        got = p.grad.clone()
        if name == "item_embedding.weight":
            got[0] = 0                                             # padding_idx = 0 (nextitnet.py:29)
        assert (got - want).abs().max().item() <= 2e-6 * max(1.0, want.abs().max().item()), name
    with torch.no_grad():
        p = golden_params(case)
        scores = NO.predict(p, torch.from_numpy(G["item_seq"]), p["item_embedding.weight"], DIL)
        assert (scores - torch.from_numpy(G[case + "/scores"]).double()).abs().max().item() < 5e-6
