"""GRU4Rec (pixelrec_amd/model/gru4rec.py) against the reference's own outputs (tests/golden/gru4rec_tiny.npz, written by
oracle/make_golden_gru4rec.py from REC.model.IDNet.gru4rec.GRU4Rec) and, at a wider shape with the lazy table optimizer,
against torch.optim.AdamW on the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "gru4rec_tiny.npz"))
N, E, MULT, NL, L, B = [int(x) for x in G["dims"]]


def _config(e=E, mult=MULT, nl=NL, l=L):
    return {"embedding_size": e, "hidden_size": mult, "num_layers": nl, "dropout_prob": 0.0, "MAX_ITEM_LIST_LENGTH": l,
            "seed": 2020}


def test_gru4rec_matches_the_reference():
    from pixelrec_amd.model import GRU4Rec

    class DL:
        item_num = N

    m = GRU4Rec(_config(), DL())
    ref = {k[6:]: torch.from_numpy(G[k]) for k in G.files if k.startswith("param/")}
    assert list(m.state_dict().keys()) == list(ref.keys())                     # the reference's names AND order
    m.load_state_dict(ref, strict=True)
    m = m.cuda().train()
    items, mask = torch.from_numpy(G["items"]).cuda(), torch.from_numpy(G["masked_index"]).cuda()
    loss = m((items, mask))
    loss.backward()
    assert abs(float(loss.detach()) - float(G["loss"])) < 5e-6
    for name, p in m.named_parameters():
        want = torch.from_numpy(G["grad/" + name])
        if name == "item_embedding.weight":                                     # the table gradient is sparse: densify it
            sp = m.sparse_table_grad
            n = int(sp.n)
            got = torch.zeros_like(want)
            got[sp.idx[:n].cpu()] = sp.rows[:n].cpu()
        else:
            got = p.grad.cpu()
        err = (got - want).abs().max().item()
        assert err <= 2e-6 + 2e-5 * want.abs().max().item(), (name, err)
    m.eval()
    with torch.no_grad():
        scores = m.predict(torch.from_numpy(G["item_seq"]).cuda(), m.compute_item_all())
        assert (scores.cpu() - torch.from_numpy(G["scores"])).abs().max().item() < 2e-5
        assert abs(float(m((items, mask))) - float(G["loss"])) < 5e-6


@pytest.mark.parametrize("table_update", ["lazy", "dense"])
def test_gru4rec_training_steps_follow_torch_adamw_on_the_oracle(table_update):
    """E = 64, H = 128, one layer, L = 10, 500 items: four optimizer steps of PxrAdamW (sparse table gradient, lazy or dense
    table schedule) against torch.optim.AdamW over the oracle (dense table gradient with row 0 zeroed)."""
    from oracle import gru4rec_oracle as GO
    from pixelrec_amd.model import GRU4Rec
    from pixelrec_amd.optim import PxrAdamW

    n, e, mult, nl, l, b = 500, 64, 2, 1, 10, 8
    rng = np.random.default_rng(4)

    class DL:
        item_num = n

    torch.manual_seed(2)
    m = GRU4Rec(_config(e, mult, nl, l), DL())
    ref = {k: v.detach().clone().double().requires_grad_(True) for k, v in m.state_dict().items()}
    m = m.cuda().train()
    opt = PxrAdamW(m, lr=1e-3, weight_decay=0.1, table_update=table_update)
    topt = torch.optim.AdamW(list(ref.values()), lr=1e-3, weight_decay=0.1)
    for step in range(4):
        items = torch.from_numpy(rng.integers(1, n, size=(b, 2, l + 1)).astype(np.int64))
        mask = torch.ones(b, l, dtype=torch.int64)
        items[0, 0, :3] = 0; mask[0, :3] = 0
        loss = m((items.cuda(), mask.cuda()))
        loss.backward()
        opt.step()
        topt.zero_grad()
        rl = GO.forward_loss(ref, items, mask, nl)
        rl.backward()
        ref["item_embedding.weight"].grad[0] = 0                                # padding_idx = 0
        topt.step()
        assert abs(float(loss.detach()) - float(rl.detach())) < 2e-5 * max(1.0, abs(float(rl.detach()))), step
    sd = m.state_dict()
    for k, v in ref.items():
        err = (sd[k].cpu().double() - v.detach()).abs().max().item()
        assert err < 4e-5, (k, err)                                             # 4 steps of lr 1e-3 at the 1e-5-per-step budget


def test_gru4rec_embedding_dropout_matches_the_oracle_with_the_library_mask():
    """dropout_prob > 0 (reference gru4rec.py:26,59): loss and gradients of a training step against the oracle with the kernels'
    counter-hash keep mask injected (oracle/dropout_rng.py; mask index = the time-major [L, B, E] element); evaluation never drops;
    a second step draws a different mask (the completed-step counter is part of the seed)."""
    from oracle import dropout_rng
    from pixelrec_amd.model import GRU4Rec

    class DL:
        item_num = N

    from oracle import gru4rec_oracle as GO

    p_drop = 0.3
    cfg = dict(_config(), dropout_prob=p_drop)
    m = GRU4Rec(cfg, DL())
    ref = {k[6:]: torch.from_numpy(G[k]).clone() for k in G.files if k.startswith("param/")}
    m.load_state_dict(ref, strict=True)
    m = m.cuda().train()
    items, mask = torch.from_numpy(G["items"]), torch.from_numpy(G["masked_index"])
    B = items.shape[0]
    losses = []
    for step in range(2):
        seed = (m._emb_drop_seed() + step) & 0xFFFFFFFFFFFFFFFF
        keep_tm = dropout_rng.keep_mask(seed, m.EMB_DROP_STREAM, (L, B, E), p_drop)             # time-major, as the kernel counts
        keep = torch.from_numpy(np.ascontiguousarray(keep_tm.transpose(1, 0, 2)))
        leaf = {k: v.clone().requires_grad_(True) for k, v in ref.items()}
        loss_ref = GO.forward_loss(leaf, items, mask.float(), NL, emb_keep=keep, p_drop=p_drop)
        loss_ref.backward()
        loss = m((items.cuda(), mask.cuda()))
        loss.backward()
        losses.append(float(loss))
        assert abs(float(loss) - float(loss_ref)) <= 3e-5 * max(1.0, abs(float(loss_ref)))
        for name, prm in m.named_parameters():
            g_ref = leaf[name].grad
            if name == "item_embedding.weight":                                 # sparse table gradient: densify; padding row zeroed
                sp = m.sparse_table_grad
                n = int(sp.n)
                got = torch.zeros_like(g_ref)
                got[sp.idx[:n].cpu()] = sp.rows[:n].cpu()
                g_ref = g_ref.clone(); g_ref[0].zero_()
            else:
                got = prm.grad.cpu()
            assert (got - g_ref).abs().max().item() <= 5e-6 + 2e-4 * g_ref.abs().max().item(), (step, name)
    assert losses[0] != losses[1]
    m.eval()
    with torch.no_grad():                                                       # evaluation never drops
        l_eval = float(m((items.cuda(), mask.cuda())))
    assert abs(l_eval - float(GO.forward_loss(ref, items, mask.float(), NL))) <= 3e-5 * max(1.0, abs(l_eval))
    assert abs(l_eval - float(G["loss"])) < 5e-6
    with pytest.raises(ValueError):
        m((torch.zeros(2, 2, L, dtype=torch.int64).cuda(), torch.ones(2, L, dtype=torch.int64).cuda()))


@pytest.mark.parametrize("fused", [True, False])
def test_trainer_runs_gru4rec_end_to_end(tmp_path, fused):
    """IDNet/gru4rec.yaml-shaped run on TinyInter: Trainer.fit (hipGraph replay of the step), full-sort evaluation, a
    checkpoint with the reference's keys whose optimizer entry loads into a torch AdamW over the reference's parameter order."""
    from pixelrec_amd.config import Config
    from pixelrec_amd.data import bulid_dataloader, load_data
    from pixelrec_amd.optim import reference_rec_parameter_names
    from pixelrec_amd.parallel import DataParallel
    from pixelrec_amd.trainer import Trainer
    from pixelrec_amd.utils import get_model

    golden_dir = os.path.join(os.path.dirname(__file__), "golden")
    my, ov = tmp_path / "m.yaml", tmp_path / "o.yaml"
    my.write_text("model: GRU4Rec\nembedding_size: 32\nhidden_size: 1\nnum_layers: 1\ndropout_prob: 0\n")
    ov.write_text(f"seed: 2020\nstate: INFO\nuse_modality: False\nreproducibility: True\ncheckpoint_dir: '{tmp_path}/saved'\n"
                  f"log_path: '{tmp_path}/log'\nshow_progress: False\nMAX_ITEM_LIST_LENGTH: 6\ndata_path: {golden_dir}/\n"
                  "dataset: TinyInter\nepochs: 3\ntrain_batch_size: 8\noptim_args: {learning_rate: 0.003, weight_decay: 0.1}\n"
                  "eval_batch_size: 16\ntopk: [5,10]\nmetrics: ['Recall', 'NDCG']\nvalid_metric: NDCG@10\n"
                  f"metric_decimal_place: 7\neval_step: 1\nstopping_step: 30\neval_fused_topk: {fused}\n")
    config = Config([str(my), str(ov)])
    config["device"] = torch.device("cuda", 0)
    dataload = load_data(config)
    train, valid, test = bulid_dataloader(config, dataload)
    model = get_model(config["model"])(config, dataload)
    trainer = Trainer(config, DataParallel(model.to(config["device"])))
    trainer.fit(train, valid, saved=True)
    losses = [trainer.train_loss_dict[e] for e in sorted(trainer.train_loss_dict)]
    assert len(losses) == 3 and losses[-1] < losses[0]
    res = trainer.evaluate(test, load_best_model=True)
    assert set(res) == {"recall@5", "recall@10", "ndcg@5", "ndcg@10"}
    ck = torch.load(trainer.saved_model_file, map_location="cpu", weights_only=False)
    names = reference_rec_parameter_names(trainer.model.module)
    assert names == list(ck["state_dict"].keys()) == ["item_embedding.weight", "gru_layers.weight_ih_l0",
                                                      "gru_layers.weight_hh_l0", "dense.weight", "dense.bias"]
    tparams = [torch.nn.Parameter(ck["state_dict"][k].clone()) for k in names]
    topt = torch.optim.AdamW(tparams, lr=1.0, weight_decay=0.5)
    topt.load_state_dict(ck["optimizer"])
    assert topt.param_groups[0]["lr"] == 0.003 and topt.state[tparams[1]]["exp_avg"].shape == tparams[1].shape
