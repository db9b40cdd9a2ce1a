"""CPU ORACLE for the PixelNet path -- TEST INFRASTRUCTURE ONLY.

Pinned against the reference's own classes: tests/golden/mosasrec_tiny.npz is produced by oracle/make_golden_pixel.py
from the REFERENCE's `MOSASRec` (code/REC/model/PixelNet/mosasrec.py:66-119) and `MeanItemEncoder`
(code/REC/model/layers.py:121-128), imported unmodified, over a random tiny HF `CLIPVisionModel`;
tests/test_mosasrec_golden.py checks this restatement (and the HIP path) against it: loss, all 53 gradients,
`compute_item`, `predict`.  What stays UNPINNED: the reference's `load_model` ViT branch itself cannot run offline
(load.py:94 downloads `openai/clip-vit-base-patch32`; load.py:112,116 poke `model.vision_model`, which transformers
5.x no longer has -- SURVEY.md §7 hard part 8), so the goldens use the installed transformers 5.x tower, not the
reference's pinned 4.16.2, and random rather than pre-trained weights.  The sequence part re-uses
oracle.sasrec_oracle, which is pinned against the reference's SASRec.
"""
from __future__ import annotations

import torch

from . import sasrec_oracle as O


def hf_clip_vision(hidden, n_layers, heads, inter, image_size, patch_size):
    from transformers import CLIPVisionConfig, CLIPVisionModel

    cfg = CLIPVisionConfig(hidden_size=hidden, intermediate_size=inter, num_hidden_layers=n_layers,
                           num_attention_heads=heads, image_size=image_size, patch_size=patch_size)
    return CLIPVisionModel(cfg).eval()


def hf_state_to_reference_names(hf_model):
    """transformers 5.x names -> the 4.16.2 names the reference checkpoints carry (`vision_model.` prefix)."""
    return {"vision_model." + k: v.detach().clone() for k, v in hf_model.state_dict().items()}


def mean_item_encoder(hf_model, rec_w, rec_b, images):
    """MeanItemEncoder.forward (layers.py:125-128) with rec_fc = Linear + ReLU: [n,3,H,W] -> [n,D]."""
    x = hf_model(pixel_values=images)[0]                    # last_hidden_state (before post_layernorm)
    return torch.relu(x @ rec_w.t() + rec_b).mean(dim=1)


def forward_loss(seq_params, item_emb, masked_index, cfg):
    """MOSASRec.forward after the encoder (mosasrec.py:69-93): item_emb [B, L+1, 2, D] (pos|neg interleaved)."""
    pos, neg = item_emb[:, :, 0], item_emb[:, :, 1]
    inp, tp, tn = pos[:, :-1], pos[:, 1:], neg[:, 1:]
    L = masked_index.shape[1]
    x = inp + seq_params["position_embedding.weight"][:L][None]
    h = O.layer_norm(x, seq_params["LayerNorm.weight"], seq_params["LayerNorm.bias"], cfg["layer_norm_eps"])
    mask = O.attention_mask(masked_index)
    for i in range(cfg["n_layers"]):
        h = O.encoder_layer(seq_params, i, h, mask, cfg["n_heads"], cfg["layer_norm_eps"])
    ps, ns = (h * tp).sum(-1), (h * tn).sum(-1)
    loss = -(torch.log((ps - ns).sigmoid() + 1e-8) * masked_index).sum(-1)
    return loss.mean(-1)
