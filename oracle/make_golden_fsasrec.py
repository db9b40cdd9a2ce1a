"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/fsasrec_tiny.npz by running the REFERENCE's FSASRec class
(code/REC/model/ViNet/fsasrec.py) with its own load_weights / FIX / HY / SEMATIC item encoders, unmodified.

Run in the dev container (where /root/reference exists):   python oracle/make_golden_fsasrec.py

For each of freeze_model / hybrid_model / semantic_model (the three sasrec_v / sasrec_vid / sasrec_semantic_id YAMLs of
code/ViNet): a tiny feature (or PQ-code) file, the model built with a fixed seed, dropout probabilities 0 (so that the
training-mode forward is deterministic), and stored: the inputs, the state_dict (reference key names), the training loss,
the gradient of every parameter, compute_item_all() and predict() scores.  SURVEY.md §8 f4.
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402

N_ITEMS, F, D, L, H, NL, B, C, CODE_CAP = 30, 12, 16, 5, 2, 2, 3, 4, 7


def main():
    # resolve transformers' lazy modules BEFORE the inert torchvision stub exists (REC/model/load.py:3,5 imports these names)
    from transformers import BeitModel, CLIPVisionModel, SwinConfig, SwinModel, ViTMAEModel  # noqa: F401

    ref_shim.import_reference()
    from REC.model.ViNet.fsasrec import FSASRec

    rng = np.random.default_rng(11)
    feats = rng.standard_normal((N_ITEMS, F)).astype(np.float32)
    codes = rng.integers(0, CODE_CAP, size=(N_ITEMS, C)).astype(np.int64)
    codes[3, 0] = CODE_CAP - 1                     # the maximum of column 0 defines code_cap (layers.py:216)
    codes[0, :] = 0                                # item 0 (padding) -> code row 0 = padding_idx in its first position
    tmp = tempfile.mkdtemp()
    fpath, cpath = os.path.join(tmp, "feat.npy"), os.path.join(tmp, "codes.npy")
    np.save(fpath, feats); np.save(cpath, codes)

    items = rng.integers(1, N_ITEMS, size=(B, 2, L + 1)).astype(np.int64)
    mask = np.ones((B, L), dtype=np.int64)
    items[0, 0, :2] = 0; mask[0, :1] = 0           # left padding, as SEQTrainDataset produces it
    items[1, 0, :3] = 0; mask[1, :2] = 0
    item_seq = items[:, 0, 1:].copy()

    class DL:
        item_num = N_ITEMS

    out = {"feats": feats, "codes": codes, "items": items, "masked_index": mask, "item_seq": item_seq,
           "dims": np.array([N_ITEMS, F, D, L, H, NL, B, C])}
    for kind, flags, dnn in (("fix", {"freeze_model": True}, []), ("fixmlp", {"freeze_model": True}, [20]),
                             ("hybrid", {"hybrid_model": True}, []), ("semantic", {"semantic_model": True}, [])):
        config = {"n_layers": NL, "n_heads": H, "embedding_size": D, "inner_size": 2, "hidden_dropout_prob": 0.0,
                  "attn_dropout_prob": 0.0, "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02,
                  "MAX_ITEM_LIST_LENGTH": L, "device": "cpu", "v_feat_path": fpath, "semantic_id_path": cpath,
                  "dnn_layers": dnn, "freeze_model": False, "hybrid_model": False, "semantic_model": False, **flags}
        torch.manual_seed(5)
        m = FSASRec(config, DL())
        # weights at initializer_range 0.02 make every ReLU unit of the projection a coin flip around 0 and the loss flat;
        # scale the encoder up so that gradients are well away from rounding (values are stored, nothing depends on the init)
        with torch.no_grad():
            for n_, p in m.named_parameters():
                if n_.startswith("item_embedding."):
                    p.mul_(8.0)
                if n_.endswith("bias") or "LayerNorm" in n_:
                    p.add_(0.05 * torch.randn_like(p))
        m.train()
        loss = m((torch.from_numpy(items), torch.from_numpy(mask)))
        loss.backward()
        for k, v in m.state_dict().items():
            out[f"{kind}/param/{k}"] = v.detach().numpy().copy()
        for n_, p in m.named_parameters():
            out[f"{kind}/grad/{n_}"] = (p.grad if p.grad is not None else torch.zeros_like(p)).detach().numpy().copy()
        out[f"{kind}/loss"] = np.array(float(loss.detach()))
        m.eval()
        with torch.no_grad():
            feat_all = m.compute_item_all()
            out[f"{kind}/item_all"] = feat_all.numpy().copy()
            out[f"{kind}/scores"] = m.predict(torch.from_numpy(item_seq), feat_all).numpy().copy()
        print(kind, "loss", float(loss.detach()), "params", sum(p.numel() for p in m.parameters()))
    dst = os.path.join(ROOT, "tests", "golden", "fsasrec_tiny.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, os.path.getsize(dst), "bytes")


if __name__ == "__main__":
    main()
