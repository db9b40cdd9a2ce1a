"""Host profile of one Trainer epoch at a given batch size on the Pixel200K-shaped synthetic dataset (where does an epoch's wall time go
when it is not the device?).   TT_BATCH=512 python tools/diag/trainer_epoch_profile.py"""
import cProfile
import os
import pstats
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import run as R  # noqa: E402
import synth_dataset  # noqa: E402

B = int(os.environ.get("TT_BATCH", "512"))
tmp = tempfile.mkdtemp(prefix="pxr_tp_")
synth_dataset.main(os.path.join(tmp, "data"), 200_000, 96_000)
cfg = dict(model="SASRec", n_layers=2, n_heads=4, embedding_size=512, inner_size=2, hidden_dropout_prob=0.1, attn_dropout_prob=0.1,
           hidden_act="gelu", layer_norm_eps=1e-12, initializer_range=0.02, seed=2020, state="WARNING", use_modality=False,
           reproducibility=True, checkpoint_dir=os.path.join(tmp, "saved"), log_path=os.path.join(tmp, "log"), show_progress=False,
           MAX_ITEM_LIST_LENGTH=50, data_path=os.path.join(tmp, "data") + "/", dataset="Pixel200K", epochs=3, train_batch_size=B,
           optim_args={"learning_rate": 1e-4, "weight_decay": 0.1}, eval_batch_size=1024, topk=[5, 10], metrics=["Recall", "NDCG"],
           valid_metric="NDCG@10", metric_decimal_place=7, eval_step=1, stopping_step=30)
config, dataload, (train, valid, test), model = R.build(0, config_dict=cfg)
from pixelrec_amd.trainer import Trainer  # noqa: E402

tr = Trainer(config, model)
tr._train_epoch(train, 0)
torch.cuda.synchronize()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
tr._train_epoch(train, 1)
torch.cuda.synchronize()
pr.disable()
dt = time.perf_counter() - t0
print(f"B={B}: epoch {dt:.2f} s, {len(train)} steps, {dt / len(train) * 1e3:.2f} ms per step, graph {tr._gstep is not None}")
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
