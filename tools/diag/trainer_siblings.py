"""The sibling backbones (GRU4Rec, NextItNet with the values of their reference YAMLs) end to end through the Trainer on the
Pixel200K-shaped synthetic dataset: two epochs + one evaluation each -- does the loop run and is anything pathological on the host?
python tools/diag/trainer_siblings.py"""
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

tmp = tempfile.mkdtemp(prefix="pxr_sib_")
synth_dataset.main(os.path.join(tmp, "data"), 100_000, 96_000)
base = dict(seed=2020, state="WARNING", use_modality=False, reproducibility=True, checkpoint_dir=os.path.join(tmp, "saved"),
            log_path=os.path.join(tmp, "log"), show_progress=False, MAX_ITEM_LIST_LENGTH=10, data_path=os.path.join(tmp, "data") + "/",
            dataset="Pixel200K", epochs=2, train_batch_size=64, optim_args={"learning_rate": 1e-4, "weight_decay": 0.1},
            eval_batch_size=1024, topk=[5, 10], metrics=["Recall", "NDCG"], valid_metric="NDCG@10", metric_decimal_place=7, eval_step=1,
            stopping_step=30)
models = {"GRU4Rec": dict(model="GRU4Rec", embedding_size=2048, hidden_size=1, num_layers=1, dropout_prob=0),
          "NextItNet": dict(model="NextItNet", embedding_size=1024, kernel_size=3, block_num=3, dilations=[1, 4], final_layer=False),
          "SASRec": dict(model="SASRec", n_layers=2, n_heads=4, embedding_size=512, inner_size=2, hidden_dropout_prob=0.1,
                         attn_dropout_prob=0.1, hidden_act="gelu", layer_norm_eps=1e-12, initializer_range=0.02)}
from pixelrec_amd.trainer import Trainer  # noqa: E402

for name, mc in models.items():
    config, dataload, (train, valid, test), model = R.build(0, config_dict={**base, **mc})
    tr = Trainer(config, model)
    for ep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        loss = tr._train_epoch(train, ep)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"{name}: epoch {ep} {dt:.2f} s, {len(train)} steps, {dt / len(train) * 1e3:.3f} ms/step, {len(train) * 64 / dt / 1e3:.1f} K sequences/s, "
              f"loss {loss:.1f}, graph {tr._gstep is not None}", flush=True)
    t0 = time.perf_counter()
    res = tr.evaluate(valid, load_best_model=False)
    print(f"{name}: evaluation of 100000 users {time.perf_counter() - t0:.2f} s ndcg@10 {res['ndcg@10']}", flush=True)
    del tr, model
    torch.cuda.empty_cache()
