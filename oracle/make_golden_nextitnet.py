"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/nextitnet_tiny.npz by running the REFERENCE's NextItNet class
(code/REC/model/IDNet/nextitnet.py) unmodified, with and without its final_layer: inputs, state_dict, training loss, the
gradient of every parameter (the table's dense, row 0 zero through padding_idx), predict scores.

Run in the dev container (where /root/reference exists):   python oracle/make_golden_nextitnet.py
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402

N_ITEMS, E, K, BLOCKS, L, B = 40, 16, 3, 2, 7, 4
DIL = [1, 2]


def main():
    from transformers import BeitModel, CLIPVisionModel, SwinConfig, SwinModel, ViTMAEModel  # noqa: F401  (before the stubs)

    ref_shim.import_reference()
    from REC.model.IDNet.nextitnet import NextItNet

    rng = np.random.default_rng(31)
    items = rng.integers(1, N_ITEMS, size=(B, 2, L + 1)).astype(np.int64)
    mask = np.ones((B, L), dtype=np.int64)
    items[0, 0, :2] = 0; mask[0, :2] = 0
    items[3, 0, :5] = 0; mask[3, :5] = 0
    item_seq = items[:, 0, 1:].copy()
    out = {"items": items, "masked_index": mask, "item_seq": item_seq, "dims": np.array([N_ITEMS, E, K, BLOCKS, L, B]),
           "dilations": np.array(DIL)}

    class DL:
        item_num = N_ITEMS
        user_num = 7

    for case, final in (("plain", False), ("final", True)):
        config = {"embedding_size": E, "block_num": BLOCKS, "dilations": DIL, "kernel_size": K, "reg_weight": 0.0,
                  "final_layer": final}
        torch.manual_seed(13)
        m = NextItNet(config, DL())
        with torch.no_grad():      # the reference's table init is U(+-1/sqrt(N)) = +-0.16: keep; move LayerNorm / bias values off 1 / 0
            for n_, p in m.named_parameters():
                if ".ln" in n_ or n_.endswith("bias"):
                    p.add_(0.1 * torch.randn_like(p))
        m.train()
        loss = m((torch.from_numpy(items), torch.from_numpy(mask)))
        loss.backward()
        out[f"{case}/loss"] = np.array(float(loss.detach()))
        for k_, v in m.state_dict().items():
            out[f"{case}/param/{k_}"] = v.detach().numpy().copy()
        for n_, p in m.named_parameters():
            out[f"{case}/grad/{n_}"] = p.grad.detach().numpy().copy()
        m.eval()
        with torch.no_grad():
            out[f"{case}/scores"] = m.predict(torch.from_numpy(item_seq), m.compute_item_all()).numpy().copy()
        print(case, "loss", float(loss.detach()), "params", len(list(m.parameters())))
    dst = os.path.join(ROOT, "tests", "golden", "nextitnet_tiny.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst))


if __name__ == "__main__":
    main()
