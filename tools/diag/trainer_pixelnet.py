"""End-to-end PixelNet epoch through the Trainer on a small synthetic dataset with a synthetic image store (use_modality: the images are
assembled on the device from the HBM-resident store; CLIP ViT-B/16 with random weights, the last two blocks train): does the whole
loop -- batcher, image batches, tower forward / backward, both optimizer groups, item-feature pass, full-sort evaluation -- run, and how
long do its phases take?   python tools/diag/trainer_pixelnet.py [n_users] [n_items]"""
import os
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import run as R  # noqa: E402
import synth_dataset  # noqa: E402

n_users = int(sys.argv[1]) if len(sys.argv) > 1 else 1600
n_items = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
tmp = tempfile.mkdtemp(prefix="pxr_tp_")
synth_dataset.main(os.path.join(tmp, "data"), n_users, n_items)
cfg = dict(model="MOSASRec", n_layers=2, n_heads=4, embedding_size=512, inner_size=2, hidden_dropout_prob=0.1, attn_dropout_prob=0.1,
           hidden_act="gelu", layer_norm_eps=1e-12, initializer_range=0.02, seed=2020, state="INFO", use_modality=True,
           reproducibility=True, checkpoint_dir=os.path.join(tmp, "saved"), log_path=os.path.join(tmp, "log"), show_progress=False,
           MAX_ITEM_LIST_LENGTH=10, data_path=os.path.join(tmp, "data") + "/", dataset="Pixel200K", image_path="synthetic:224",
           encoder_name="clip-vit-base-patch16", encoder_source="transformers",
           fine_tune_arg={"tune_scale": 165, "pre_trained": False, "allow_random_backbone": True, "activation": "relu", "dnn_layers": [],
                          "method": "mean"},
           epochs=2, train_batch_size=16, optim_args={"modal_lr": 1e-4, "rec_lr": 1e-4, "modal_decay": 0, "rec_decay": 0.1},
           eval_batch_size=1024, topk=[5, 10], metrics=["Recall", "NDCG"], valid_metric="NDCG@10", metric_decimal_place=7, eval_step=1,
           stopping_step=30)
t0 = time.perf_counter()
config, dataload, (train, valid, test), model = R.build(0, config_dict=cfg)
print(f"build: {time.perf_counter() - t0:.1f} s; items {dataload.item_num}, train steps per epoch {len(train)}")
from pixelrec_amd.trainer import Trainer  # noqa: E402

tr = Trainer(config, model)
for ep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = tr._train_epoch(train, ep)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"epoch {ep}: {dt:.2f} s, {len(train)} steps, {dt / len(train) * 1e3:.1f} ms/step, {len(train) * 16 / dt:.0f} sequences/s, loss {loss}")
t0 = time.perf_counter()
res = tr.evaluate(valid, load_best_model=False)
torch.cuda.synchronize()
print(f"evaluation (item features of {dataload.item_num} images + full sort): {time.perf_counter() - t0:.2f} s  {dict(res)}")
