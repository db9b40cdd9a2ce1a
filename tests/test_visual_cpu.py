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
        out_hf = hf(pixel_values=x)
        ref, got = out_hf[0], mine(x)[0]
        # [1] = pooler_output = post_layernorm(class token): what fine_tune_arg.method 'pool' feeds rec_fc (layers.py:130-137)
        ref_pool, got_pool = out_hf[1], mine(x)[1]
    assert (ref - got).abs().max().item() < 2e-5
    assert (ref_pool - got_pool).abs().max().item() < 2e-5


def test_pool_method_keeps_post_layernorm_a_trainable_parameter():
    """load.py:119-120: the 'pool' branch does not swap post_layernorm for Identity; at a tune_scale below its index it trains."""
    from pixelrec_amd.model import visual

    cfg = {"encoder_name": "clip-vit-tiny-test", "encoder_source": "transformers", "embedding_size": 24, "pretrain_path": None,
           "fine_tune_arg": {"tune_scale": 5 + 16, "pre_trained": False, "activation": "relu", "dnn_layers": [], "method": "pool"}}
    enc = visual.load_model(cfg)
    names = [n for n, p in enc.named_parameters() if p.requires_grad]
    assert "item_encoder.vision_model.post_layernorm.weight" in names and "item_encoder.vision_model.post_layernorm.bias" in names
    x = torch.randn(2, 3, 64, 64)
    out = enc(x)                                                            # CPU restatement
    last, pooled = enc.item_encoder(x)
    assert out.shape == (2, 24) and torch.equal(out, enc.rec_fc(pooled))
    assert torch.allclose(pooled, torch.nn.functional.layer_norm(last[:, 0, :], (last.shape[-1],),
                                                                 enc.item_encoder.vision_model.post_layernorm.weight,
                                                                 enc.item_encoder.vision_model.post_layernorm.bias, 1e-5))
    for m in ("mean", "cls"):                                               # ... the other methods still drop it (load.py:112,116)
        cfg["fine_tune_arg"]["method"] = m
        assert not any("post_layernorm" in n for n, _ in visual.load_model(cfg).named_parameters())


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
        with pytest.raises(RuntimeError, match="available offline"):
            visual.load_model(cfg)
        # pre_trained: False loads the same checkpoint in the reference (load.py:101) -- it needs the weights just as much
        cfg["fine_tune_arg"] = dict(cfg["fine_tune_arg"], pre_trained=False)
        with pytest.raises(RuntimeError, match="available offline"):
            visual.load_model(cfg)
        cfg["fine_tune_arg"] = dict(cfg["fine_tune_arg"], pre_trained=True, allow_random_backbone=True)
        enc = visual.load_model(cfg)               # explicit opt-in: builds (with a logged warning)
        assert sum(not p.requires_grad for p in enc.item_encoder.parameters()) == 165
        assert not any("post_layernorm" in n for n, _ in enc.named_parameters())       # Identity (load.py:112,116)
    finally:
        visual.CLIPVisionEncoder = saved
    assert len(built) == 3


def test_pre_trained_false_keeps_the_frozen_front_and_reinitialises_the_tail(tmp_path, monkeypatch):
    """load.py:101-108: with `pre_trained: False` the reference STILL starts from the pre-trained checkpoint and re-draws only
    the trainable tail (index >= tune_scale) from N(0, 0.02).  Round trip: a tiny HF `CLIPVisionModel` saved with
    save_pretrained() is picked up through `encoder_path`."""
    from transformers import CLIPVisionConfig, CLIPVisionModel

    from pixelrec_amd.model import visual

    hidden, layers, heads, inter, image, patch = visual.ENCODER_SHAPES["clip-vit-tiny-test"]
    torch.manual_seed(3)
    hf = CLIPVisionModel(CLIPVisionConfig(hidden_size=hidden, num_hidden_layers=layers, num_attention_heads=heads,
                                          intermediate_size=inter, image_size=image, patch_size=patch))
    hf.save_pretrained(tmp_path / "tiny_clip")
    monkeypatch.setitem(visual.ENCODER_SHAPES, "clip-vit-tiny-disk", visual.ENCODER_SHAPES["clip-vit-tiny-test"])
    ref = {("vision_model." + k if not k.startswith("vision_model.") else k): v for k, v in hf.state_dict().items()}
    tune = 5 + 16 * 2                                # embeddings + pre-LN + blocks 0, 1 frozen; block 2 trains
    for pre in (True, False):
        cfg = {"encoder_name": "clip-vit-tiny-disk", "encoder_source": "transformers", "embedding_size": 24, "pretrain_path": None,
               "encoder_path": str(tmp_path / "tiny_clip"),
               "fine_tune_arg": {"tune_scale": tune, "pre_trained": pre, "activation": "relu", "dnn_layers": [], "method": "mean"}}
        enc = visual.load_model(cfg)
        named = list(enc.item_encoder.named_parameters())
        assert len(named) == len(ref) - 2 and not any("post_layernorm" in n for n, _ in named)
        for i, (n, p) in enumerate(named):
            assert p.requires_grad == (i >= tune), n
            same = torch.equal(p.detach(), ref[n])
            if i < tune or pre:
                assert same, (pre, n)                # the checkpoint's values: frozen front always, everything with pre_trained
            else:
                assert not same and abs(float(p.std()) - 0.02) < 0.01, (n, float(p.std()))   # re-drawn N(0, 0.02)
