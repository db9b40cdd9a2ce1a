"""The restated CLIP vision tower == HF CLIPVisionModel on identical weights (CPU; the torch fallback of the patch
projection is exercised here, the MFMA path in the GPU tests)."""
import pytest
import torch

from oracle import mosasrec_oracle as MO


def test_clip_tower_matches_hf_names_and_outputs():
    from pixelrec_amd.model.visual import CLIPVisionEncoder, ENCODER_SHAPES

    shape = ENCODER_SHAPES["clip-vit-tiny-test"]
    torch.manual_seed(0)
    hf = MO.hf_clip_vision(*shape)
    mine = CLIPVisionEncoder(*shape).eval()
    sd = MO.hf_state_to_reference_names(hf)
    assert [k for k, _ in mine.named_parameters()] == list(sd.keys())      # same names AND order (tune_scale indexes it)
    mine.load_state_dict(sd, strict=True)
    x = torch.randn(3, 3, 64, 64)
    with torch.no_grad():
        ref = hf(pixel_values=x)[0]
        got = mine(x)[0]
    assert (ref - got).abs().max().item() < 2e-5


def test_b32_parameter_layout_matches_reference_tune_scale():
    from pixelrec_amd.model.visual import CLIPVisionEncoder, ENCODER_SHAPES

    with torch.device("meta"):
        m = CLIPVisionEncoder(*ENCODER_SHAPES["clip-vit-base-patch32"])
    names = [k for k, _ in m.named_parameters()]
    assert len(names) == 199                                               # 5 + 16*12 + 2
    assert names[165] == "vision_model.encoder.layers.10.self_attn.k_proj.weight"
    assert names[1] == "vision_model.embeddings.patch_embedding.weight"


def test_pre_trained_without_local_weights_fails_loudly(tmp_path):
    """overall/ViT.yaml ships `pre_trained: True`; the reference downloads the CLIP weights (load.py:94).  Offline there is
    nothing to load: training on a frozen random backbone must be an explicit choice, not a silent default."""
    from pixelrec_amd.model import visual

    built = []
    real = visual.CLIPVisionEncoder

    def meta_encoder(*shape):                      # keep the test cheap: parameters on the meta device
        with torch.device("meta"):
            m = real(*shape)
        built.append(m)
        return m

    visual.CLIPVisionEncoder, saved = meta_encoder, real
    try:
        cfg = {"encoder_name": "clip-vit-base-patch32", "encoder_source": "transformers", "embedding_size": 32,
               "pretrain_path": None, "encoder_path": str(tmp_path / "no_such_checkpoint"),
               "fine_tune_arg": {"tune_scale": 165, "pre_trained": True, "activation": "relu", "dnn_layers": [],
                                 "method": "mean"}}
        with pytest.raises(RuntimeError, match="pre_trained"):
            visual.load_model(cfg)
        cfg["fine_tune_arg"]["allow_random_backbone"] = True
        enc = visual.load_model(cfg)               # explicit opt-in: builds (with a logged warning)
        assert sum(not p.requires_grad for p in enc.item_encoder.parameters()) == 165
        cfg["fine_tune_arg"] = dict(cfg["fine_tune_arg"], pre_trained=False, allow_random_backbone=False)
    finally:
        visual.CLIPVisionEncoder = saved
    assert len(built) == 2
