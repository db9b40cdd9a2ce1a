"""Ten epochs + evaluations of the Trainer on the Pixel200K-shaped synthetic dataset: device memory (allocated / reserved), host RSS and
epoch time after every epoch -- nothing may grow.   python tools/diag/trainer_soak.py [epochs]"""
import os
import resource
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import run as R  # noqa: E402
import synth_dataset  # noqa: E402

epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 10
tmp = tempfile.mkdtemp(prefix="pxr_soak_")
synth_dataset.main(os.path.join(tmp, "data"), 200_000, 96_000)
cfg = dict(model="SASRec", n_layers=2, n_heads=4, embedding_size=512, inner_size=2, hidden_dropout_prob=0.1, attn_dropout_prob=0.1,
           hidden_act="gelu", layer_norm_eps=1e-12, initializer_range=0.02, seed=2020, state="WARNING", use_modality=False,
           reproducibility=True, checkpoint_dir=os.path.join(tmp, "saved"), log_path=os.path.join(tmp, "log"), show_progress=False,
           MAX_ITEM_LIST_LENGTH=50, data_path=os.path.join(tmp, "data") + "/", dataset="Pixel200K", epochs=epochs, train_batch_size=64,
           optim_args={"learning_rate": 1e-4, "weight_decay": 0.1}, eval_batch_size=1024, topk=[5, 10], metrics=["Recall", "NDCG"],
           valid_metric="NDCG@10", metric_decimal_place=7, eval_step=1, stopping_step=30)
config, dataload, (train, valid, test), model = R.build(0, config_dict=cfg)
from pixelrec_amd.trainer import Trainer  # noqa: E402

tr = Trainer(config, model)
for ep in range(epochs):
    train.sampler.set_epoch(ep) if hasattr(train, "sampler") and hasattr(train.sampler, "set_epoch") else None
    t0 = time.perf_counter()
    loss = tr._train_epoch(train, ep)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    res = tr.evaluate(valid, load_best_model=False)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"epoch {ep}: train {t1 - t0:.2f} s  eval {t2 - t1:.2f} s  loss {loss:.1f}  ndcg@10 {res['ndcg@10']:.5f}  "
          f"device allocated {torch.cuda.memory_allocated() / 2**20:.0f} MiB reserved {torch.cuda.memory_reserved() / 2**20:.0f} MiB  "
          f"host max RSS {resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1024:.0f} MiB", flush=True)
