"""End-to-end epoch throughput of the Trainer (host batcher + prefetcher + captured step) on a Pixel200K-SHAPED synthetic dataset
(tools/synth_dataset.py), against bench.py's step on batches that are already resident: is the host side keeping up?
python tools/diag/trainer_throughput.py [n_users] [n_items]"""
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

n_users = int(sys.argv[1]) if len(sys.argv) > 1 else 200_000
n_items = int(sys.argv[2]) if len(sys.argv) > 2 else 96_000
tmp = tempfile.mkdtemp(prefix="pxr_tt_")
t0 = time.perf_counter()
synth_dataset.main(os.path.join(tmp, "data"), n_users, n_items)
print(f"dataset written in {time.perf_counter() - t0:.1f} s")
cfg = dict(model="SASRec", n_layers=2, n_heads=4, embedding_size=512, inner_size=2, hidden_dropout_prob=0.1, attn_dropout_prob=0.1,
           hidden_act="gelu", layer_norm_eps=1e-12, initializer_range=0.02, seed=2020, state="INFO", use_modality=False,
           reproducibility=True, checkpoint_dir=os.path.join(tmp, "saved"), log_path=os.path.join(tmp, "log"), show_progress=False,
           MAX_ITEM_LIST_LENGTH=50, data_path=os.path.join(tmp, "data") + "/", dataset="Pixel200K", epochs=3,
           train_batch_size=int(os.environ.get("TT_BATCH", "64")),
           optim_args={"learning_rate": 1e-4, "weight_decay": 0.1}, eval_batch_size=1024, topk=[5, 10], metrics=["Recall", "NDCG"],
           valid_metric="NDCG@10", metric_decimal_place=7, eval_step=1, stopping_step=30)
t0 = time.perf_counter()
config, dataload, (train, valid, test), model = R.build(0, config_dict=cfg)
print(f"build (load + index the CSV, loaders, model): {time.perf_counter() - t0:.1f} s; items {dataload.item_num}, train steps per epoch {len(train)}")
from pixelrec_amd.trainer import Trainer  # noqa: E402

tr = Trainer(config, model)
for ep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    loss = tr._train_epoch(train, ep)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    n_seq = len(train) * cfg["train_batch_size"]
    print(f"epoch {ep}: {dt:.2f} s, {len(train)} steps, {dt / len(train) * 1e3:.3f} ms/step, {n_seq / dt / 1e3:.1f} K sequences/s, loss {loss}")
t0 = time.perf_counter()
res = tr.evaluate(valid, load_best_model=False)
torch.cuda.synchronize()
print(f"full-sort evaluation of {len(valid.dataset) if hasattr(valid, 'dataset') else '?'} users: {time.perf_counter() - t0:.2f} s  {dict(res)}")

# where the evaluation's time goes (host profile of one more pass)
import cProfile  # noqa: E402
import pstats  # noqa: E402

pr = cProfile.Profile()
pr.enable()
tr.evaluate(valid, load_best_model=False)
torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)

# the whole fit loop: 3 epochs, validation after each, checkpoints when the metric improves
config2, dataload2, (train2, valid2, test2), model2 = R.build(0, config_dict=cfg)
tr2 = Trainer(config2, model2)
t0 = time.perf_counter()
best, best_res = tr2.fit(train2, valid2, saved=True)
torch.cuda.synchronize()
print(f"fit (3 epochs + 3 validations + checkpoints): {time.perf_counter() - t0:.1f} s; best {best}")
t0 = time.perf_counter()
res = tr2.evaluate(test2, load_best_model=True)
print(f"test evaluation incl. loading the best checkpoint: {time.perf_counter() - t0:.1f} s {dict(res)}")
