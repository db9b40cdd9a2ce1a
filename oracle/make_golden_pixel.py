"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/mosasrec_tiny.npz by running the REFERENCE's MOSASRec class
(code/REC/model/PixelNet/mosasrec.py) and its MeanItemEncoder (code/REC/model/layers.py:121-128) unmodified.

Run in the dev container (where /root/reference exists):   python oracle/make_golden_pixel.py

The reference's `load_model` (load.py:90-120) cannot run offline: it downloads `openai/clip-vit-base-patch32`.  It is
replaced -- in the imported module's namespace only -- by a function that does what load.py:91-120 does around the
download: build an HF `CLIPVisionModel` (here random-init, tiny config, installed transformers), freeze the first
`tune_scale` named parameters, wrap it in the reference's own `MeanItemEncoder`.  Everything downstream -- the
pos/neg interleave `.view(B,-1,2,D)`, the sequence block, the loss, `predict`, `compute_item` -- is the reference's
code.  Stored: inputs, the state_dict (reference key names), and the reference's outputs.  SURVEY.md §8c G6.
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402

SHAPE = (64, 3, 4, 128, 64, 32)      # hidden, layers, heads, mlp, image, patch  (= ENCODER_SHAPES["clip-vit-tiny-test"])
D, L, H, NL, B, N_ITEMS = 64, 6, 2, 2, 3, 40
TUNE = 5 + 16 * 2                    # freeze embeddings + blocks 0, 1; train block 2


def main(dnn_layers=(), out_name="mosasrec_tiny.npz", store_images=True):
    """dnn_layers = () is the shipped head (one Linear + ReLU, overall/ViT.yaml:33); a non-empty list selects the reference's
    MLPLayers head (layers.py:69-71, 239-294) -- that golden reuses the images of the first one (same generator seed) and does not
    store them again."""
    # resolve transformers' lazy modules BEFORE the inert torchvision stub exists (its availability probe needs a real
    # module spec); these are the names REC/model/load.py:3,5 imports
    from transformers import BeitModel, CLIPVisionModel, SwinConfig, SwinModel, ViTMAEModel  # noqa: F401

    ref_shim.import_reference()
    import REC.model.PixelNet.mosasrec as ref_mod
    from REC.model.layers import MeanItemEncoder
    from transformers import CLIPVisionConfig, CLIPVisionModel

    def load_model(config):
        hidden, n_layers, heads, inter, image, patch = SHAPE
        cfg = CLIPVisionConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=n_layers,
                               num_attention_heads=heads, image_size=image, patch_size=patch)
        model = CLIPVisionModel(cfg)
        for index, (_, p) in enumerate(model.named_parameters()):       # load.py:97-99
            if index < config["fine_tune_arg"]["tune_scale"]:
                p.requires_grad = False
        return MeanItemEncoder(item_encoder=model, input_dim=hidden, output_dim=config["embedding_size"],
                               act_name="relu", dnn_layers=list(dnn_layers))          # load.py:116-117

    ref_mod.load_model = load_model
    config = {"n_layers": NL, "n_heads": H, "embedding_size": D, "inner_size": 2, "hidden_dropout_prob": 0.1,
              "attn_dropout_prob": 0.1, "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02,
              "MAX_ITEM_LIST_LENGTH": L, "pretrain_path": None,
              "fine_tune_arg": {"tune_scale": TUNE, "pre_trained": True, "activation": "relu", "dnn_layers": list(dnn_layers),
                                "method": "mean"}}

    class DL:
        item_num = N_ITEMS

    torch.manual_seed(2020)
    model = ref_mod.MOSASRec(config, DL())
    with torch.no_grad():                                               # make biases / LN affine non-trivial
        for n, p in model.named_parameters():
            if n.endswith("bias") or "LayerNorm" in n or "layer_norm" in n or "layrnorm" in n:
                p.add_(0.05 * torch.randn_like(p))
    model.eval()                                                        # dropout off

    g = torch.Generator().manual_seed(7)
    images = torch.randn(B, 2 * (L + 1), 3, SHAPE[4], SHAPE[4], generator=g)
    mask = torch.ones(B, L, dtype=torch.int64)
    mask[0, :3] = 0
    images[0, :8] = 0.0                                                 # left-padded row: zero images (trainset.py:137)
    loss = model((images, mask))
    loss.backward()

    store = {"meta": np.array([D, L, H, NL, B, N_ITEMS, TUNE]), "shape": np.array(SHAPE),
             "images": images.numpy().astype(np.float16),               # the test feeds exactly these (fp16-exact) values
             "masked_index": mask.numpy()}
    # re-run on the fp16-rounded images so that the stored inputs reproduce the stored outputs exactly
    images = torch.from_numpy(store["images"].astype(np.float32))
    model.zero_grad()
    loss = model((images, mask))
    loss.backward()
    store["loss"] = loss.detach().numpy()
    for k, v in model.state_dict().items():
        store["param." + k] = v.numpy()
    n_grads = 0
    for k, p in model.named_parameters():
        if p.grad is not None:
            store["grad." + k] = p.grad.numpy()
            n_grads += 1
    store["frozen"] = np.array([k for k, p in model.named_parameters() if not p.requires_grad])

    item_imgs = torch.randn(N_ITEMS, 3, SHAPE[4], SHAPE[4], generator=g).half().float()
    item_imgs[0] = 0.0
    feat = model.compute_item(item_imgs)
    item_seq = torch.randint(1, N_ITEMS, (4, L), generator=g)
    item_seq[0, :2] = 0
    scores = model.predict(item_seq, feat)
    store["eval.item_images"] = item_imgs.numpy().astype(np.float16)
    if not store_images:         # identical to the arrays of mosasrec_tiny.npz (checked by the test that reads both)
        store.pop("images")
        store.pop("eval.item_images")
        store["dnn_layers"] = np.array(list(dnn_layers))
    store["eval.item_feature"] = feat.numpy()
    store["eval.item_seq"] = item_seq.numpy()
    store["eval.scores"] = scores.numpy()

    out = os.path.join(ROOT, "tests", "golden", out_name)
    np.savez_compressed(out, **store)
    print(f"wrote {out}: {os.path.getsize(out) / 1e6:.2f} MB, {n_grads} gradients, loss {float(loss):.6f}")


if __name__ == "__main__":
    main()
    main(dnn_layers=(48,), out_name="mosasrec_dnn_tiny.npz", store_images=False)
