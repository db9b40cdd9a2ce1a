"""Two data-parallel ranks sharing cuda:0 (gloo transport; the 8-GPU node uses RCCL through the same code): the
full product path incl. the HIP merge of the all-gathered sparse table gradients.  Invariant: one DP step over
2 ranks x B sequences == one single-process oracle step on the concatenated 2B batch (mean of the per-rank means
== global mean for equal per-rank batches, which is DDP's averaging convention)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu

META = dict(n_items=400, D=64, L=10, H=2, inner=2, n_layers=2)
B = 6


def _batches():
    from pixelrec_amd import synth

    rng = np.random.default_rng(5)
    z = synth.ZipfItems(META["n_items"], seed=1)
    return [synth.train_batch(META["n_items"], B, META["L"], rng, z) for _ in range(2 * 2)]   # 2 steps x 2 ranks


def _worker(rank, port, results):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE="2")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=2)
    try:
        from oracle import sasrec_oracle as O
        from pixelrec_amd.model import SASRec
        from pixelrec_amd.optim import PxrAdamW
        from pixelrec_amd.parallel import DataParallel

        cfg = {"n_layers": 2, "n_heads": 2, "embedding_size": 64, "inner_size": 2, "hidden_dropout_prob": 0.0,
               "attn_dropout_prob": 0.0, "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02,
               "MAX_ITEM_LIST_LENGTH": 10, "seed": 2020}

        class DL:
            item_num = META["n_items"]

        params = O.synth_params(META["n_items"], 64, 10, 2, 2, seed=9)
        model = SASRec(cfg, DL())
        if rank == 0:
            model.load_state_dict(params, strict=True)       # rank 1 keeps its own random init: the broadcast fixes it
        dp = DataParallel(model.cuda().train())
        opt = PxrAdamW(model, lr=1e-3, weight_decay=0.1)
        batches = _batches()
        for step in range(2):
            it, mk = batches[2 * step + rank]
            loss = dp((torch.from_numpy(it).cuda(), torch.from_numpy(mk).cuda()))
            loss.backward()
            dp.sync_gradients()
            opt.step()
        sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
        if rank == 0:
            tr = O.OracleTrainer(params, {"n_layers": 2, "n_heads": 2, "layer_norm_eps": 1e-12}, lr=1e-3, weight_decay=0.1)
            for step in range(2):
                it = np.concatenate([batches[2 * step][0], batches[2 * step + 1][0]])
                mk = np.concatenate([batches[2 * step][1], batches[2 * step + 1][1]])
                tr.step(torch.from_numpy(it), torch.from_numpy(mk))
            worst = max((sd[k] - tr.p[k]).abs().max().item() for k in sd)
            results["err_vs_oracle"] = worst
        results[f"sum{rank}"] = float(sum(v.double().sum() for v in sd.values()))
        results[f"tab{rank}"] = sd["item_embedding.weight"].numpy().tobytes()
    finally:
        dist.destroy_process_group()


def test_two_ranks_equal_one_big_batch():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    with mp.Manager() as mgr:
        results = mgr.dict()
        mp.spawn(_worker, args=(port, results), nprocs=2, join=True)
        r = dict(results)
    assert r["err_vs_oracle"] < 2e-5, r["err_vs_oracle"]
    assert r["tab0"] == r["tab1"]                      # replicas stay bit-identical
    assert r["sum0"] == r["sum1"]
