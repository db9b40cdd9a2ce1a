"""Visual item encoder for the PixelNet models -- the ViT branch of the reference's `load_model`
(code/REC/model/load.py:90-120) plus the item-encoder heads of code/REC/model/layers.py:65-128.

The reference builds HF `CLIPVisionModel.from_pretrained('openai/clip-vit-base-patch32')`, freezes the first
`tune_scale` named parameters (165 => only encoder blocks 10 and 11 train), replaces `post_layernorm` by Identity and
wraps it as `MeanItemEncoder`: out = mean over the 50 tokens of ReLU(rec_fc(last_hidden_state)).  Pretrained weights
cannot be downloaded here (no network) and the installed transformers (5.x) no longer has the `.vision_model`
attribute the reference pokes (SURVEY.md §7 hard part 8), so the encoder is restated as a small torch module with the
SAME parameter names/order as transformers 4.16.2's `CLIPVisionModel` (`vision_model.embeddings.class_embedding`, ...,
199 named parameters for B/32; index 165 = first parameter of block 10): reference checkpoints load by name.

What runs where: on a HIP tensor the WHOLE encoder -- patch projection, every ViT block forward, the backward of the
trainable blocks, rec_fc + ReLU + token mean -- runs on this build's kernels (model/vit_native.py); the torch modules in
this file hold the parameters under the reference's names and restate the tower for CPU-side checks (tests/test_visual_cpu.py).
"""
from __future__ import annotations


import torch
import torch.nn as nn

from .. import ops


class _Embeddings(nn.Module):
    def __init__(self, hidden, image_size, patch_size):
        super().__init__()
        self.patch_size = patch_size
        self.num_patches = (image_size // patch_size) ** 2
        self.class_embedding = nn.Parameter(torch.randn(hidden))
        self.patch_embedding = nn.Conv2d(3, hidden, kernel_size=patch_size, stride=patch_size, bias=False)
        self.position_embedding = nn.Embedding(self.num_patches + 1, hidden)
        self.use_hip_gemm = True

    def forward(self, pixel_values):
        n, c, H, W = pixel_values.shape
        p = self.patch_size
        w = self.patch_embedding.weight
        if self.use_hip_gemm and pixel_values.is_cuda and not (torch.is_grad_enabled() and w.requires_grad):
            # im2col is a pure layout change; the contraction runs on the fp32 MFMA GEMM (exact fp32)
            patches = pixel_values.view(n, c, H // p, p, W // p, p).permute(0, 2, 4, 1, 3, 5).reshape(-1, c * p * p)
            x = ops.linear_fwd(patches.contiguous(), w.detach().view(w.shape[0], -1).contiguous(), None)
            x = x.view(n, self.num_patches, -1)
        else:  # trainable patch projection (tune_scale < 2) or CPU construction-time checks
            x = self.patch_embedding(pixel_values).flatten(2).transpose(1, 2)
        cls = self.class_embedding.expand(n, 1, -1)
        x = torch.cat([cls, x], dim=1)
        return x + self.position_embedding.weight[None]


class _Attention(nn.Module):
    def __init__(self, hidden, heads):
        super().__init__()
        self.heads, self.d = heads, hidden // heads
        self.k_proj = nn.Linear(hidden, hidden)
        self.v_proj = nn.Linear(hidden, hidden)
        self.q_proj = nn.Linear(hidden, hidden)
        self.out_proj = nn.Linear(hidden, hidden)

    def forward(self, x):
        n, t, hdim = x.shape
        q = self.q_proj(x) * (self.d ** -0.5)
        k, v = self.k_proj(x), self.v_proj(x)
        sh = lambda z: z.view(n, t, self.heads, self.d).transpose(1, 2)
        att = torch.softmax(sh(q) @ sh(k).transpose(-1, -2), dim=-1)
        return self.out_proj((att @ sh(v)).transpose(1, 2).reshape(n, t, hdim))


class _MLP(nn.Module):
    def __init__(self, hidden, inter):
        super().__init__()
        self.fc1 = nn.Linear(hidden, inter)
        self.fc2 = nn.Linear(inter, hidden)

    def forward(self, x):
        x = self.fc1(x)
        return self.fc2(x * torch.sigmoid(1.702 * x))   # quick_gelu


class _EncoderLayer(nn.Module):
    def __init__(self, hidden, heads, inter, eps):
        super().__init__()
        self.self_attn = _Attention(hidden, heads)
        self.layer_norm1 = nn.LayerNorm(hidden, eps=eps)
        self.mlp = _MLP(hidden, inter)
        self.layer_norm2 = nn.LayerNorm(hidden, eps=eps)

    def forward(self, x):
        x = x + self.self_attn(self.layer_norm1(x))
        return x + self.mlp(self.layer_norm2(x))


class _Encoder(nn.Module):
    def __init__(self, n_layers, hidden, heads, inter, eps):
        super().__init__()
        self.layers = nn.ModuleList([_EncoderLayer(hidden, heads, inter, eps) for _ in range(n_layers)])


class _VisionTransformer(nn.Module):
    def __init__(self, hidden, n_layers, heads, inter, image_size, patch_size, eps=1e-5):
        super().__init__()
        self.embeddings = _Embeddings(hidden, image_size, patch_size)
        self.pre_layrnorm = nn.LayerNorm(hidden, eps=eps)      # (sic) the HF attribute name
        self.encoder = _Encoder(n_layers, hidden, heads, inter, eps)
        self.post_layernorm = nn.LayerNorm(hidden, eps=eps)    # load_model swaps it for Identity once the freeze indices
                                                               # are assigned, as the reference does (load.py:112,116)

    def forward(self, pixel_values):
        x = self.pre_layrnorm(self.embeddings(pixel_values))
        for layer in self.encoder.layers:
            x = layer(x)
        return x                                              # last_hidden_state


class CLIPVisionEncoder(nn.Module):
    """`CLIPVisionModel` look-alike: parameters are named `vision_model.*` like transformers 4.16.2."""

    def __init__(self, hidden=768, n_layers=12, heads=12, inter=3072, image_size=224, patch_size=32):
        super().__init__()
        self.vision_model = _VisionTransformer(hidden, n_layers, heads, inter, image_size, patch_size)
        self.hidden = hidden

    def forward(self, pixel_values):
        # (last_hidden_state, pooler_output) like HF: pooler_output = post_layernorm(class token); with post_layernorm swapped for
        # Identity (methods 'mean' / 'cls', load.py:112,116) it is the class token itself and nobody reads it
        last = self.vision_model(pixel_values)
        return (last, self.vision_model.post_layernorm(last[:, 0, :]))


ENCODER_SHAPES = {
    # name -> (hidden, layers, heads, mlp, image, patch)
    "clip-vit-base-patch32": (768, 12, 12, 3072, 224, 32),     # the encoder the reference ships (overall/ViT.yaml)
    "clip-vit-base-patch16": (768, 12, 12, 3072, 224, 16),     # BASELINE configs[2]
    "clip-vit-large-patch14": (1024, 24, 16, 4096, 224, 14),   # BASELINE configs[4]
    "clip-vit-tiny-test": (64, 3, 4, 128, 64, 32),     # unit tests only
    "clip-vit-tiny64-test": (128, 3, 2, 256, 64, 16),  # unit tests only: head size 64 (the fused tower attention / fp16 planes path)
}


def _activation(act_name):
    """activation_layer (layers.py:23-49) for the names the head supports; 'none' / None -> no module at all in an MLP head."""
    key = act_name.lower() if isinstance(act_name, str) else act_name
    table = {"relu": nn.ReLU, "sigmoid": nn.Sigmoid, "tanh": nn.Tanh, None: None, "none": None}
    if key not in table:
        raise NotImplementedError(f"activation {act_name!r}")
    return table[key]() if table[key] is not None else None


class MLPHead(nn.Module):
    """The head `fine_tune_arg.dnn_layers: [h1, h2, ...]` selects (reference MLPLayers, layers.py:239-294, built by
    PatchItemEncoder, layers.py:69-71, with dropout 0 and bn=False): for every consecutive pair of [input_dim, h1, ..., output_dim]
    a Dropout(0), a Linear and -- unless the activation is 'none' -- the activation, ALSO after the last Linear.  The container
    is `mlp_layers` and the modules keep their positions in it, so the parameter names are the reference's
    (`rec_fc.mlp_layers.1.weight`, `rec_fc.mlp_layers.4.weight`, ...: checkpoints interchange)."""

    def __init__(self, sizes, act_name):
        super().__init__()
        mods = []
        for fan_in, fan_out in zip(sizes[:-1], sizes[1:]):
            mods += [nn.Dropout(p=0.0), nn.Linear(fan_in, fan_out)]
            a = _activation(act_name)
            if a is not None:
                mods.append(a)
        self.mlp_layers = nn.Sequential(*mods)

    def forward(self, x):
        return self.mlp_layers(x)


class _ItemEncoderBase(nn.Module):
    """PatchItemEncoder (layers.py:65-92): backbone + rec_fc = Linear(input_dim, output_dim) + activation -- or, with
    `dnn_layers`, the MLP head above -- xavier-normal weights / zero biases (layers.py:78-86)."""

    def __init__(self, item_encoder, input_dim, output_dim, act_name="relu", dnn_layers=None):
        super().__init__()
        _activation(act_name)        # unsupported names raise here
        self.item_encoder = item_encoder
        # the MI355X-native forward/backward of the whole encoder (model/vit_native.py) used for every HIP tensor; the
        # torch modules below are parameter containers (reference names) and the CPU restatement used by CPU-only checks
        from .vit_native import NativeTower

        self._native = NativeTower(self)
        self._anchor = None
        # weights rewritten behind the tower's back (load_state_dict; the trainable blocks never use cached planes)
        self.register_load_state_dict_post_hook(lambda mod, _keys: mod._native.drop_weight_planes())
        if dnn_layers:
            self.rec_fc = MLPHead([input_dim] + [int(h) for h in dnn_layers] + [output_dim], act_name)
        else:
            self.rec_fc = nn.Sequential(nn.Linear(input_dim, output_dim), _activation(act_name) or nn.Identity())
        for mod in self.rec_fc.modules():
            if isinstance(mod, nn.Linear):
                nn.init.xavier_normal_(mod.weight.data)
                nn.init.constant_(mod.bias.data, 0)

    def head_layers(self):
        """[(weight name, bias name, activation module | None)] of the head's Linear layers in order (names relative to this module)."""
        if isinstance(self.rec_fc, MLPHead):
            mods = list(self.rec_fc.mlp_layers)
            out = []
            for i, mod in enumerate(mods):
                if isinstance(mod, nn.Linear):
                    nxt = mods[i + 1] if i + 1 < len(mods) and not isinstance(mods[i + 1], (nn.Linear, nn.Dropout)) else None
                    out.append((f"rec_fc.mlp_layers.{i}.weight", f"rec_fc.mlp_layers.{i}.bias", nxt))
            return out
        a = self.rec_fc[1]
        return [("rec_fc.0.weight", "rec_fc.0.bias", None if isinstance(a, nn.Identity) else a)]


class MeanItemEncoder(_ItemEncoderBase):
    native_method = "mean"

    def forward(self, x):                       # layers.py:125-128
        if x.is_cuda:
            from . import vit_native

            return vit_native.run(self, x)
        x = self.item_encoder(x)[0]
        return torch.mean(self.rec_fc(x), dim=1)


class ClsItemEncoder(_ItemEncoderBase):
    native_method = "cls"

    def forward(self, x):                       # layers.py:113-117
        if x.is_cuda:
            from . import vit_native

            return vit_native.run(self, x)
        x = self.item_encoder(x)[0]
        return self.rec_fc(x[:, 0, :])


class PoolItemEncoder(_ItemEncoderBase):
    """rec_fc on HF's `pooler_output` = post_layernorm(class token) (layers.py:130-137; load.py:119-120 keeps post_layernorm a
    parameter of the model for this method -- indices 197, 198 of CLIP ViT-B: trainable at the shipped tune_scale)."""
    native_method = "pool"

    def forward(self, x):
        if x.is_cuda:
            from . import vit_native

            return vit_native.run(self, x)
        return self.rec_fc(self.item_encoder(x)[1])


def _load_pretrained_backbone(model, name, config):
    """fine_tune_arg.pre_trained (the shipped default, overall/ViT.yaml): the reference calls
    CLIPVisionModel.from_pretrained('openai/<name>') (load.py:94).  Same here, restricted to what is on local disk (no
    network): `encoder_path` (a directory / checkpoint in HF format) or the local HF cache.  Returns an error string
    when nothing could be loaded."""
    path = config["encoder_path"] if "encoder_path" in config else None
    try:
        from transformers import CLIPVisionModel

        hf = CLIPVisionModel.from_pretrained(path or f"openai/{name}", local_files_only=True)
    except Exception as e:  # noqa: BLE001 - any failure means "no local weights"
        return f"{type(e).__name__}: {e}"
    sd = {(k if k.startswith("vision_model.") else "vision_model." + k): v for k, v in hf.state_dict().items()}
    model.load_state_dict(sd, strict=True)      # transformers 4.16.2 names == this module's names
    return None


def load_model(config):
    """ViT branch of the reference's load_model (load.py:90-120).  The reference ALWAYS starts from the pre-trained
    checkpoint (`CLIPVisionModel.from_pretrained` in both branches, load.py:94,101) and, with `pre_trained: False`,
    re-initialises only the TRAINABLE tail (index >= tune_scale) with N(0, 0.02) (load.py:104-108) -- the frozen front keeps
    the pre-trained weights either way.  Same here: the backbone weights must come from local disk (`encoder_path` / the HF
    cache) or from `pretrain_path` (a Trainer checkpoint loaded by the model afterwards); otherwise this raises instead of
    training on a frozen RANDOM backbone without saying so (`allow_random_backbone: True` opts out).  `post_layernorm` is
    replaced by Identity AFTER the freeze indices are assigned (load.py:112,116): it counts for `tune_scale` but is not a
    parameter of the model, its optimizer group or its checkpoints."""
    name, source = config["encoder_name"], config["encoder_source"]
    if source != "transformers" or name not in ENCODER_SHAPES:
        raise NotImplementedError(f"visual encoder {source}/{name} is outside this build's scope "
                                  f"(built: transformers/{sorted(ENCODER_SHAPES)})")
    ft = config["fine_tune_arg"] or {}
    tune_scale = ft.get("tune_scale", 0)
    pre_trained = ft.get("pre_trained", True)
    model = CLIPVisionEncoder(*ENCODER_SHAPES[name])
    if not name.endswith("-test"):
        err = _load_pretrained_backbone(model, name, config)
        if err is not None and not config["pretrain_path"]:
            if not ft.get("allow_random_backbone", False):
                raise RuntimeError(
                    f"no weights for 'openai/{name}' are available offline ({err}); the reference loads them in both "
                    "fine_tune_arg.pre_trained branches (load.py:94,101).  Point `encoder_path` at a local copy of the HF "
                    "checkpoint, give `pretrain_path`, or set fine_tune_arg.allow_random_backbone: True.")
            import logging

            logging.getLogger().warning("visual encoder %s: NO pre-trained weights loaded (%s); the first %d parameters "
                                        "stay frozen at their RANDOM initial values", name, err, tune_scale)
    for index, (pname, param) in enumerate(model.named_parameters()):
        if index < tune_scale:
            param.requires_grad = False                                 # load.py:97-99
        elif not pre_trained:
            param.data.normal_(mean=0.0, std=0.02)                      # load.py:104-108
    method = ft.get("method", "mean")
    cls = {"mean": MeanItemEncoder, "cls": ClsItemEncoder, "pool": PoolItemEncoder}.get(method)
    if cls is None:
        raise NotImplementedError(f"fine_tune_arg.method={method!r} (built: mean, cls, pool)")
    if method != "pool":
        model.vision_model.post_layernorm = nn.Identity()               # load.py:112,116 (the 'pool' branch, :119-120, keeps it)
    return cls(model, model.hidden, config["embedding_size"], ft.get("activation", "relu"), ft.get("dnn_layers"))
