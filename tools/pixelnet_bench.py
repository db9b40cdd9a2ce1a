"""PixelNet (MOSASRec + CLIP ViT, end-to-end image encoder) training-step timing on one GPU, synthetic images, random
init, fp32, the reference's shipped batch shape (train_batch_size 16, MAX_ITEM_LIST_LENGTH 10 => 352 images/step).
usage: python tools/pixelnet_bench.py [encoder_name] [tune_from_block] [batch] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from pixelrec_amd.model import MOSASRec
from pixelrec_amd.model.visual import ENCODER_SHAPES
from pixelrec_amd.optim import OptimizerGroup, PxrAdamW, VisualAdamW

name = sys.argv[1] if len(sys.argv) > 1 else "clip-vit-base-patch32"
hidden, n_layers, heads, inter, image, patch = ENCODER_SHAPES[name]
tune_from = int(sys.argv[2]) if len(sys.argv) > 2 else n_layers - 2       # reference: tune_scale 165 = blocks 10, 11 of 12
B = int(sys.argv[3]) if len(sys.argv) > 3 else 16
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 10
L, D = 10, 512
cfg = {"n_layers": 2, "n_heads": 4, "embedding_size": D, "inner_size": 2, "hidden_dropout_prob": 0.1,
       "attn_dropout_prob": 0.1, "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02,
       "MAX_ITEM_LIST_LENGTH": L, "seed": 2020, "encoder_name": name, "encoder_source": "transformers",
       "pretrain_path": None,
       "fine_tune_arg": {"tune_scale": 5 + 16 * tune_from, "pre_trained": True, "allow_random_backbone": True, "activation": "relu", "dnn_layers": [],
                         "method": "mean"}}


class DL:
    item_num = 96001


torch.manual_seed(0)
m = MOSASRec(cfg, DL()).cuda().train()
opt = OptimizerGroup(VisualAdamW(m.visual_encoder, lr=1e-4, weight_decay=0.0), PxrAdamW(m, lr=1e-4, weight_decay=0.1))
images = torch.randn(B, 2 * (L + 1), 3, image, image, device="cuda")
mask = torch.ones(B, L, dtype=torch.int64, device="cuda")


def step():
    opt.zero_grad()
    loss = m((images, mask))
    loss.backward()
    opt.step()
    return loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    loss = step()
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / steps
n_img = B * 2 * (L + 1)
tokens = n_img * ((image // patch) ** 2 + 1)
fwd = tokens * n_layers * (24 * hidden * hidden + 4 * ((image // patch) ** 2 + 1) * hidden) \
    + n_img * ((image // patch) ** 2) * 2 * 3 * patch * patch * hidden
trained = (n_layers - tune_from) / n_layers
print(f"{name}: B={B} ({n_img} images, {tokens} tokens) {dt * 1e3:.1f} ms/step  {B / dt:.1f} sequences/s  "
      f"{n_img / dt:.0f} images/s  ~{fwd * (1 + 2 * trained) / dt / 1e12:.1f} TFLOP/s (fp32, fwd + bwd of {n_layers - tune_from} blocks)  "
      f"loss {float(loss):.4f}")
