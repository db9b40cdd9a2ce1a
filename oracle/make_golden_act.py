"""TEST INFRASTRUCTURE ONLY -- tests/golden/sasrec_act.npz: the REFERENCE SASRec (imported unmodified through
oracle/ref_shim.py) with each non-default `hidden_act` of its FeedForward (layers.py:642-649: relu / swish / tanh /
sigmoid): loss, three gradients and the eval scores of one tiny seeded case per activation.
Run in the dev container:   python oracle/make_golden_act.py"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402
from oracle.sasrec_oracle import synth_params  # noqa: E402
from pixelrec_amd import synth  # noqa: E402

CASE = dict(n_items=200, D=32, L=8, H=2, inner=2, n_layers=2, B=4, seed=21)
ACTS = ("relu", "swish", "tanh", "sigmoid")
GRADS = ("trm_encoder.layer.0.feed_forward.dense_1.weight", "trm_encoder.layer.1.feed_forward.dense_2.bias",
         "position_embedding.weight")


def main():
    ref_shim.import_reference()
    from REC.model.IDNet.sasrec import SASRec

    c = CASE
    rng = np.random.default_rng(c["seed"])
    items, mask = synth.train_batch(c["n_items"], c["B"], c["L"], rng, synth.ZipfItems(c["n_items"], seed=c["seed"]))
    seq = synth.eval_batch(c["n_items"], 3, c["L"], rng)[0]
    store = {"meta": np.array([c[k] for k in ("n_items", "D", "L", "H", "inner", "n_layers", "B", "seed")]),
             "items": items, "masked_index": mask, "eval.item_seq": seq}

    class DL:
        item_num = c["n_items"]

    params = synth_params(c["n_items"], c["D"], c["L"], c["n_layers"], c["inner"], seed=c["seed"])
    for act in ACTS:
        cfg = {"n_layers": c["n_layers"], "n_heads": c["H"], "embedding_size": c["D"], "inner_size": c["inner"],
               "hidden_dropout_prob": 0.1, "attn_dropout_prob": 0.1, "hidden_act": act, "layer_norm_eps": 1e-12,
               "initializer_range": 0.02, "MAX_ITEM_LIST_LENGTH": c["L"]}
        torch.manual_seed(c["seed"])
        model = SASRec(cfg, DL())
        model.load_state_dict(params, strict=True)
        model.eval()
        model.zero_grad()
        loss = model((torch.from_numpy(items), torch.from_numpy(mask)))
        loss.backward()
        store[f"{act}.loss"] = np.array(loss.item(), dtype=np.float32)
        named = dict(model.named_parameters())
        for g in GRADS:
            store[f"{act}.grad.{g}"] = named[g].grad.detach().numpy().astype(np.float32)
        with torch.no_grad():
            store[f"{act}.scores"] = model.predict(torch.from_numpy(seq), model.compute_item_all()).numpy().astype(np.float32)
    np.savez_compressed(os.path.join(ROOT, "tests", "golden", "sasrec_act.npz"), **store)
    print("wrote sasrec_act.npz", {a: float(store[f"{a}.loss"]) for a in ACTS})


if __name__ == "__main__":
    main()
