"""TEST INFRASTRUCTURE ONLY -- generates tests/golden/*.npz by running the REFERENCE itself.

Run in the dev container (where /root/reference exists):   python oracle/make_golden.py
The reference is imported unmodified through oracle/ref_shim.py (sys.modules stubs for absent third-party
packages that are not on the SASRec arithmetic path).  Only INPUTS and the reference's OUTPUTS are stored;
weights are regenerated from a seed by oracle.sasrec_oracle.synth_params (numpy PCG64), so the fixtures stay
small and contain no reference source in any form.

Cases:
  tiny  : N=300  D=64  L=12 H=2 inner=2 layers=2 B=6   everything stored, incl. per-stage activations
  cfg1  : N=1000 D=128 L=20 H=4 inner=2 layers=2 B=4   BASELINE.json configs[0] shape (small catalogue)
  ns    : N=2000 D=512 L=50 H=4 inner=2 layers=2 B=3   north-star kernel shapes (d=128), small catalogue
Each case holds: train batch -> loss, scores, last-layer states, gradients (large tensors sub-sampled +
sum / L2); eval batch -> seq_output, scores (or sub-sampled columns), masked top-10, rec.topk, recall/ndcg
sums; 4 consecutive torch.optim.AdamW steps (lr 1e-4, wd 0.1) -> parameters after each step (sub-sampled).
"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, ROOT)

from oracle import ref_shim  # noqa: E402
from oracle.sasrec_oracle import synth_params  # noqa: E402
from pixelrec_amd import synth  # noqa: E402

CASES = {
    "tiny": dict(n_items=300, D=64, L=12, H=2, inner=2, n_layers=2, B=6, seed=11),
    "cfg1": dict(n_items=1000, D=128, L=20, H=4, inner=2, n_layers=2, B=4, seed=12),
    "ns": dict(n_items=2000, D=512, L=50, H=4, inner=2, n_layers=2, B=3, seed=13),
}
SUB_LIMIT = 16384  # tensors with more elements are stored as strided rows + sum + L2


def ref_config(c):
    return {
        "n_layers": c["n_layers"], "n_heads": c["H"], "embedding_size": c["D"], "inner_size": c["inner"],
        "hidden_dropout_prob": 0.1, "attn_dropout_prob": 0.1, "hidden_act": "gelu", "layer_norm_eps": 1e-12,
        "initializer_range": 0.02, "MAX_ITEM_LIST_LENGTH": c["L"],
    }


def subsample(name, t):
    """-> dict of arrays to store for tensor t under `name`."""
    a = t.detach().cpu().numpy().astype(np.float32)
    out = {}
    if a.size <= SUB_LIMIT:
        out[name] = a
    else:
        flat = a.reshape(-1)
        stride = int(np.ceil(flat.size / SUB_LIMIT))
        out[name + "@stride"] = np.array(stride)
        out[name + "@sub"] = flat[::stride].copy()
        out[name + "@sum"] = np.array(flat.astype(np.float64).sum())
        out[name + "@l2"] = np.array(np.sqrt((flat.astype(np.float64) ** 2).sum()))
    return out


def edge_case_batch(c, rng):
    """A train batch with the edge cases the path has: a length-2 sequence (one target), left padding, a
    full-length row, an item repeated inside a sequence and across sequences, idx-0 negatives on padding."""
    items, mask = synth.train_batch(c["n_items"], c["B"], c["L"], rng, synth.ZipfItems(c["n_items"], seed=c["seed"]))
    L = c["L"]
    # row 0: shortest possible (2 items -> 1 target), everything else padding
    items[0] = 0; mask[0] = 0
    items[0, 0, -2:] = [7, 9]; items[0, 1, -1] = 5; mask[0, -1] = 1
    # row 1: full length, with a repeated item and an item shared with row 0
    items[1, 0, :] = rng.integers(1, c["n_items"], size=L + 1)
    items[1, 0, 3] = items[1, 0, 5] = 9
    items[1, 1, 1:] = rng.integers(1, c["n_items"], size=L); items[1, 1, 0] = 0
    items[1, 1, 4] = 7
    mask[1] = 1
    return items, mask


def run_case(name, c, REC):
    from REC.model.IDNet.sasrec import SASRec
    from REC.evaluator import Collector, Evaluator

    torch.manual_seed(c["seed"])
    rng = np.random.default_rng(c["seed"])

    class DL:
        item_num = c["n_items"]

    model = SASRec(ref_config(c), DL())
    params = synth_params(c["n_items"], c["D"], c["L"], c["n_layers"], c["inner"], seed=c["seed"])
    missing = model.load_state_dict(params, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    model.eval()  # dropout off; nothing else changes (no BatchNorm on this path)

    store = {"meta": np.array([c[k] for k in ("n_items", "D", "L", "H", "inner", "n_layers", "B", "seed")])}

    # ---- G1: train forward / backward -------------------------------------------------------------------
    items, mask = edge_case_batch(c, rng)
    store["items"], store["masked_index"] = items, mask
    acts = {}
    hooks = []
    hooks.append(model.LayerNorm.register_forward_hook(lambda m, i, o: acts.__setitem__("input_emb", o.detach())))
    for li, layer in enumerate(model.trm_encoder.layer):
        hooks.append(layer.multi_head_attention.register_forward_hook(
            lambda m, i, o, li=li: acts.__setitem__(f"layer{li}.attn_out", o.detach())))
        hooks.append(layer.feed_forward.register_forward_hook(
            lambda m, i, o, li=li: acts.__setitem__(f"layer{li}.ffn_out", o.detach())))
    model.zero_grad()
    loss = model((torch.from_numpy(items), torch.from_numpy(mask)))
    loss.backward()
    for h in hooks:
        h.remove()
    store["loss"] = np.array(loss.item(), dtype=np.float32)
    for k, v in acts.items():
        store.update(subsample("act." + k, v))
    # scores recomputed from the last-layer state the reference produced
    out = acts[f"layer{c['n_layers'] - 1}.ffn_out"]
    E = model.item_embedding.weight.detach()
    store["pos_score"] = (out * E[torch.from_numpy(items[:, 0, 1:])]).sum(-1).numpy()
    store["neg_score"] = (out * E[torch.from_numpy(items[:, 1, 1:])]).sum(-1).numpy()
    for k, v in model.named_parameters():
        g = v.grad
        if k == "item_embedding.weight":
            assert float(g[0].abs().max()) == 0.0  # padding_idx row
            rows = torch.nonzero(g.abs().sum(1) > 0).squeeze(1)
            store["grad.item_embedding.rows"] = rows.numpy()
            store.update(subsample("grad.item_embedding.vals", g[rows]))
            store["grad.item_embedding.rowsum"] = g.double().sum(1).numpy()
        else:
            store.update(subsample("grad." + k, g))

    # ---- G3: 4 AdamW steps on fresh batches (first one = the batch above) -------------------------------
    opt = torch.optim.AdamW(model.parameters(), lr=1e-4, weight_decay=0.1)
    zipf = synth.ZipfItems(c["n_items"], seed=c["seed"])
    steps_items, steps_mask = [items], [mask]
    for s in range(3):
        it, mk = synth.train_batch(c["n_items"], c["B"], c["L"], rng, zipf)
        steps_items.append(it); steps_mask.append(mk)
    store["adamw.items"] = np.stack(steps_items); store["adamw.masks"] = np.stack(steps_mask)
    watch_rows = np.unique(np.concatenate([[0, 1, 2, 5, 7, 9], items[:, 0].reshape(-1)[:40], [c["n_items"] - 1]]))
    store["adamw.watch_rows"] = watch_rows
    for s in range(4):
        opt.zero_grad()
        l = model((torch.from_numpy(steps_items[s]), torch.from_numpy(steps_mask[s])))
        l.backward()
        opt.step()
        store[f"adamw.loss{s}"] = np.array(l.item(), dtype=np.float32)
        sd = model.state_dict()
        store[f"adamw.step{s}.table_rows"] = sd["item_embedding.weight"][torch.from_numpy(watch_rows)].numpy()
        store[f"adamw.step{s}.table_sum"] = np.array(sd["item_embedding.weight"].double().sum().item())
        for k in ("position_embedding.weight", "LayerNorm.weight", "LayerNorm.bias",
                  "trm_encoder.layer.0.multi_head_attention.query.weight",
                  "trm_encoder.layer.1.feed_forward.dense_2.weight",
                  "trm_encoder.layer.1.feed_forward.dense_1.bias",
                  "trm_encoder.layer.0.feed_forward.LayerNorm.weight"):
            store.update(subsample(f"adamw.step{s}." + k, sd[k]))

    # ---- G2: predict + full-sort eval epilogue (on the ORIGINAL synth params again) ----------------------
    model.load_state_dict(params, strict=True)
    Be = 2 * c["B"]
    item_seq, hu, hi, pos_i = synth.eval_batch(c["n_items"], Be, c["L"], rng, zipf, hist_lo=3, hist_hi=2 * c["L"])
    store["eval.item_seq"], store["eval.history_u"], store["eval.history_i"], store["eval.positive_i"] = item_seq, hu, hi, pos_i
    scores = model.predict(torch.from_numpy(item_seq), model.compute_item_all())
    store["eval.scores"] = scores.numpy().astype(np.float32) if scores.numel() <= 1 << 18 else scores[:, ::7].numpy()
    store["eval.scores_colstride"] = np.array(1 if scores.numel() <= 1 << 18 else 7)
    masked = scores.clone()
    masked[:, 0] = -np.inf                                  # trainer.py:334
    masked[(torch.from_numpy(hu), torch.from_numpy(hi))] = -np.inf   # trainer.py:335-336
    tv, ti = torch.topk(masked, 10, dim=-1)
    store["eval.topk_idx"], store["eval.topk_val"] = ti.numpy(), tv.numpy()
    ecfg = {"metrics": ["Recall", "NDCG"], "topk": [5, 10], "device": torch.device("cpu"),
            "eval_type": None, "metric_decimal_place": 7}
    coll, ev = Collector(ecfg), Evaluator(ecfg)
    # make some users hit: plant the positive inside the top-10 for half of them
    pos_i2 = pos_i.copy()
    for u in range(0, Be, 2):
        pos_i2[u] = int(ti[u, (u // 2) % 10])
    store["eval.positive_i_planted"] = pos_i2
    coll.eval_batch_collect(masked, torch.arange(Be), torch.from_numpy(pos_i2))
    struct = coll.get_data_struct()
    store["eval.rec_topk"] = struct.get("rec.topk").numpy()
    res = ev.evaluate(struct)
    store["eval.metric_names"] = np.array(list(res.keys()))
    store["eval.metric_sums"] = np.array([float(v) for v in res.values()])

    path = os.path.join(ROOT, "tests", "golden", f"sasrec_{name}.npz")
    np.savez_compressed(path, **store)
    print(f"{name}: wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB), loss={store['loss']:.6f}")


def make_data_golden(REC):
    """G5: tiny synthetic interaction CSV -> the REFERENCE's Data.build / SeqEvalDataset / seq_eval_collate outputs."""
    import logging
    from REC.data.dataload import Data
    from REC.data.dataset import SeqEvalDataset, seq_eval_collate
    from REC.utils.enum_type import InputType

    rng = np.random.default_rng(99)
    gdir = os.path.join(ROOT, "tests", "golden")
    n_users, n_items, L = 60, 90, 6
    rows = []
    ts = rng.permutation(5000)[:2000]  # unique timestamps (pandas' default sort is not stable on ties)
    k = 0
    for u in range(n_users):
        n = int(rng.integers(3, 30))  # 3 .. 29 interactions: shorter / equal / longer than L+1 after leave-2-out
        for _ in range(n):
            rows.append((f"i{int(rng.zipf(1.3)) % n_items}", f"u{u}", int(ts[k])))
            k += 1
    order = rng.permutation(len(rows))
    csv_path = os.path.join(gdir, "TinyInter.csv")
    with open(csv_path, "w") as f:
        f.write("item_id,user_id,timestamp\n")
        for j in order:
            f.write("%s,%s,%d\n" % rows[j])
    cfg = {"data_path": gdir, "dataset": "TinyInter", "MAX_ITEM_LIST_LENGTH": L, "MODEL_INPUT_TYPE": InputType.SEQ}
    d = Data(cfg)
    d.build()
    store = {"L": np.array(L), "user_num": np.array(d.user_num), "item_num": np.array(d.item_num)}
    useq = list(d.user_seq.values())
    store["user_seq_keys"] = np.array(list(d.user_seq.keys()))
    store["user_seq_flat"] = np.concatenate(useq)
    store["user_seq_lens"] = np.array([len(x) for x in useq])
    tw = d.train_feat["item_seq"]
    store["train_flat"] = np.concatenate(tw)
    store["train_lens"] = np.array([len(x) for x in tw])
    store["train_uid"] = d.train_feat["user_id"]
    for phase in ("valid", "test"):
        ds = SeqEvalDataset(cfg, d, phase=phase)
        batch = seq_eval_collate([ds[i] for i in range(len(ds))])
        store[f"{phase}.item_seq"] = batch[0].numpy()
        store[f"{phase}.history_u"] = batch[1][0].numpy()
        store[f"{phase}.history_i"] = batch[1][1].numpy()
        store[f"{phase}.positive_u"] = batch[2].numpy()
        store[f"{phase}.item_target"] = batch[3].numpy()
    np.savez_compressed(os.path.join(gdir, "data_tiny.npz"), **store)
    print("data golden:", d.user_num, "users", d.item_num, "items", len(tw), "train windows")


def make_harness_golden(REC):
    """G5 harness level: the REFERENCE's own Config -> load_data -> bulid_dataloader -> SASRec -> DDP(gloo, CPU) ->
    Trainer.evaluate(valid/test) on TinyInter.csv with seed-generated weights -> recall/ndcg @5/@10."""
    import tempfile
    import torch.distributed as dist
    from torch.nn.parallel import DistributedDataParallel as DDP
    from REC.config import Config
    from REC.data import load_data, bulid_dataloader
    from REC.trainer import Trainer
    from REC.utils import get_model

    gdir = os.path.join(ROOT, "tests", "golden")
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
    tmp = tempfile.mkdtemp()
    model_yaml = os.path.join(tmp, "m.yaml"); over_yaml = os.path.join(tmp, "o.yaml")
    D, L, H, inner, nl = 32, 6, 2, 2, 2
    open(model_yaml, "w").write(f"model: SASRec\nn_layers: {nl}\nn_heads: {H}\nembedding_size: {D}\ninner_size: {inner}\n"
                                "hidden_dropout_prob: 0.1\nattn_dropout_prob: 0.1\nhidden_act: 'gelu'\n"
                                "layer_norm_eps: 1e-12\ninitializer_range: 0.02\n")
    open(over_yaml, "w").write(f"seed: 2020\nstate: INFO\nuse_modality: False\nreproducibility: True\n"
                               f"checkpoint_dir: '{tmp}/saved'\nshow_progress: False\nlog_wandb: False\n"
                               f"MAX_ITEM_LIST_LENGTH: {L}\ndata_path: {gdir}/\ndataset: TinyInter\nepochs: 1\n"
                               "train_batch_size: 8\noptim_args: {learning_rate: 0.0001, weight_decay: 0.1}\n"
                               "eval_batch_size: 16\ntopk: [5,10]\nmetrics: ['Recall', 'NDCG']\nvalid_metric: NDCG@10\n"
                               "metric_decimal_place: 7\neval_step: 1\nstopping_step: 30\n")
    cwd = os.getcwd(); os.chdir(tmp)
    try:
        config = Config(config_file_list=[model_yaml, over_yaml])
        config["device"] = torch.device("cpu")
        dataload = load_data(config)
        train_loader, valid_loader, test_loader = bulid_dataloader(config, dataload)
        model = get_model(config["model"])(config, dataload)
        params = synth_params(dataload.item_num, D, L, nl, inner, seed=77)
        model.load_state_dict(params, strict=True)
        ddp = DDP(model, find_unused_parameters=True)
        trainer = Trainer(config, ddp)
        store = {"meta": np.array([dataload.item_num, D, L, H, inner, nl, 77])}
        for phase, loader in (("valid", valid_loader), ("test", test_loader)):
            res = trainer.evaluate(loader, load_best_model=False)
            store[f"{phase}.names"] = np.array(list(res.keys()))
            store[f"{phase}.values"] = np.array([float(v) for v in res.values()])
            print("harness golden", phase, dict(res))
        np.savez_compressed(os.path.join(gdir, "harness_tiny.npz"), **store)
    finally:
        os.chdir(cwd)


def main():
    REC = ref_shim.import_reference()
    make_data_golden(REC)
    make_harness_golden(REC)
    os.makedirs(os.path.join(ROOT, "tests", "golden"), exist_ok=True)
    torch.set_num_threads(8)
    for name, c in CASES.items():
        run_case(name, c, REC)


if __name__ == "__main__":
    main()
