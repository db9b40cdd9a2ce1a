"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/gru4rec_tiny.npz by running the REFERENCE's GRU4Rec class
(code/REC/model/IDNet/gru4rec.py) unmodified: inputs, state_dict, training loss (dropout_prob 0, as IDNet/gru4rec.yaml
ships it), the gradient of every parameter (the table's dense, row 0 zero through padding_idx), predict scores.

Run in the dev container (where /root/reference exists):   python oracle/make_golden_gru4rec.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402

N_ITEMS, E, MULT, NL, L, B = 40, 16, 2, 2, 6, 4


def main():
    from transformers import BeitModel, CLIPVisionModel, SwinConfig, SwinModel, ViTMAEModel  # noqa: F401  (before the stubs)

    ref_shim.import_reference()
    from REC.model.IDNet.gru4rec import GRU4Rec

    rng = np.random.default_rng(21)
    items = rng.integers(1, N_ITEMS, size=(B, 2, L + 1)).astype(np.int64)
    mask = np.ones((B, L), dtype=np.int64)
    items[0, 0, :2] = 0; mask[0, :2] = 0           # left padding as SEQTrainDataset produces it (mask = input is real)
    items[2, 0, :4] = 0; mask[2, :4] = 0
    item_seq = items[:, 0, 1:].copy()
    config = {"embedding_size": E, "hidden_size": MULT, "num_layers": NL, "dropout_prob": 0.0}

    class DL:
        item_num = N_ITEMS
        user_num = 7

    torch.manual_seed(9)
    m = GRU4Rec(config, DL())
    with torch.no_grad():                          # xavier-normal rows at this size are ~0.2: fine; move the dense bias off zero
        m.dense.bias.add_(0.05 * torch.randn_like(m.dense.bias))
    m.train()
    loss = m((torch.from_numpy(items), torch.from_numpy(mask)))
    loss.backward()
    out = {"items": items, "masked_index": mask, "item_seq": item_seq, "dims": np.array([N_ITEMS, E, MULT, NL, L, B]),
           "loss": np.array(float(loss.detach()))}
    for k, v in m.state_dict().items():
        out["param/" + k] = v.detach().numpy().copy()
    for n_, p in m.named_parameters():
        out["grad/" + n_] = p.grad.detach().numpy().copy()
    m.eval()
    with torch.no_grad():
        out["scores"] = m.predict(torch.from_numpy(item_seq), m.compute_item_all()).numpy().copy()
    dst = os.path.join(ROOT, "tests", "golden", "gru4rec_tiny.npz")
    np.savez_compressed(dst, **out)
    print("loss", float(loss.detach()), "keys", list(m.state_dict().keys()), "wrote", dst, os.path.getsize(dst))


if __name__ == "__main__":
    main()
