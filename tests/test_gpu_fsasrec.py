"""FSASRec (pixelrec_amd/model/fsasrec.py) against the reference's own outputs (tests/golden/fsasrec_tiny.npz, written by
oracle/make_golden_fsasrec.py from REC.model.ViNet.fsasrec.FSASRec) and, at a wider shape, against the oracle."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "fsasrec_tiny.npz"))
N, F, D, L, H, NL, B, C = [int(x) for x in G["dims"]]
FLAGS = {"fix": ("freeze_model", []), "fixmlp": ("freeze_model", [20]), "hybrid": ("hybrid_model", []),
         "semantic": ("semantic_model", [])}


def _config(case, tmp_path, feats, codes, d=D, l=L, heads=H, p_drop=0.0):
    fpath, cpath = str(tmp_path / "feat.npy"), str(tmp_path / "codes.npy")
    np.save(fpath, feats); np.save(cpath, codes)
    flag, dnn = FLAGS[case]
    cfg = {"n_layers": NL, "n_heads": heads, "embedding_size": d, "inner_size": 2, "hidden_dropout_prob": p_drop,
           "attn_dropout_prob": p_drop, "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02,
           "MAX_ITEM_LIST_LENGTH": l, "device": "cuda", "v_feat_path": fpath, "semantic_id_path": cpath, "dnn_layers": dnn,
           "freeze_model": False, "hybrid_model": False, "semantic_model": False, "seed": 2020}
    cfg[flag] = True
    return cfg


def _grads(m):
    m.join_weight_grads()
    return {n: p.grad.detach().clone() for n, p in m.named_parameters()}


@pytest.mark.parametrize("case", list(FLAGS))
def test_fsasrec_matches_the_reference(case, tmp_path):
    from pixelrec_amd.model import FSASRec

    class DL:
        item_num = N

    m = FSASRec(_config(case, tmp_path, G["feats"], G["codes"]), DL())
    ref_params = {k[len(case) + 7:]: torch.from_numpy(G[k]) for k in G.files if k.startswith(case + "/param/")}
    assert set(ref_params) == set(m.state_dict())                  # the reference's parameter names, nothing else
    m.load_state_dict(ref_params, strict=True)
    m = m.cuda().train()
    items, mask = torch.from_numpy(G["items"]).cuda(), torch.from_numpy(G["masked_index"]).cuda()
    loss = m((items, mask))
    loss.backward()
    assert abs(float(loss.detach()) - float(G[case + "/loss"])) < 5e-6
    got = _grads(m)
    for name, g in got.items():
        want = torch.from_numpy(G[f"{case}/grad/{name}"])
        err = (g.cpu() - want).abs().max().item()
        assert err <= 2e-6 + 2e-5 * want.abs().max().item(), (name, err)
    assert any(float(torch.from_numpy(G[f"{case}/grad/{n}"]).abs().max()) > 1e-4 for n in got if n.startswith("item_embedding."))
    m.eval()
    with torch.no_grad():
        feat = m.compute_item_all()
        assert (feat.cpu() - torch.from_numpy(G[case + "/item_all"])).abs().max().item() < 5e-6
        scores = m.predict(torch.from_numpy(G["item_seq"]).cuda(), feat)
        assert (scores.cpu() - torch.from_numpy(G[case + "/scores"])).abs().max().item() < 2e-5
        # the no-grad forward is the training forward without dropout
        assert abs(float(m((items, mask))) - float(G[case + "/loss"])) < 5e-6


@pytest.mark.parametrize("case", ["fix", "hybrid", "semantic"])
def test_fsasrec_training_steps_follow_torch_adamw_on_the_oracle(case, tmp_path):
    """Wider shape (D = 64, L = 10, 300 items, 48-d features), three optimizer steps: PxrAdamW over the model's flat buffer
    against torch.optim.AdamW over the oracle's parameters (every FSASRec parameter is a 'rec' parameter, trainer.py:74-98)."""
    from oracle import fsasrec_oracle as FO
    from pixelrec_amd.model import FSASRec
    from pixelrec_amd.optim import PxrAdamW

    n, f, d, l, heads, b = 300, 48, 64, 10, 4, 6
    rng = np.random.default_rng(3)
    feats = rng.standard_normal((n, f)).astype(np.float32)
    codes = rng.integers(0, 9, size=(n, 6)).astype(np.int64)

    class DL:
        item_num = n

    torch.manual_seed(1)
    m = FSASRec(_config(case, tmp_path, feats, codes, d=d, l=l, heads=heads), DL())
    with torch.no_grad():
        for k, p in m.named_parameters():
            if k.startswith("item_embedding."):
                p.mul_(6.0)
    ref = {k: v.detach().clone().double().requires_grad_(True) for k, v in m.state_dict().items()}
    m = m.cuda().train()
    opt = PxrAdamW(m, lr=1e-3, weight_decay=0.1)
    topt = torch.optim.AdamW(list(ref.values()), lr=1e-3, weight_decay=0.1)
    kind = "fix" if case == "fix" else case
    table = FO.shifted_codes(torch.from_numpy(codes)) if case == "semantic" else torch.from_numpy(feats).double()
    cfg = {"n_layers": NL, "n_heads": heads, "layer_norm_eps": 1e-12}
    for step in range(3):
        items = torch.from_numpy(rng.integers(1, n, size=(b, 2, l + 1)).astype(np.int64))
        mask = torch.ones(b, l, dtype=torch.int64)
        items[0, 0, :3] = 0; mask[0, :2] = 0
        loss = m((items.cuda(), mask.cuda()))
        loss.backward()
        opt.step()
        topt.zero_grad()
        rl = FO.forward_loss(kind, ref, table, items, mask, cfg)
        rl.backward()
        if case == "semantic":
            ref["item_embedding.pq_code_embedding.weight"].grad[0] = 0        # padding_idx = 0
        topt.step()
        assert abs(float(loss.detach()) - float(rl.detach())) < 2e-5 * max(1.0, abs(float(rl.detach()))), step
    sd = m.state_dict()
    for k, v in ref.items():
        if k.endswith("key.bias"):
            continue      # d loss / d key.bias == 0 analytically (softmax is shift-invariant): its "gradient" is rounding noise,
            #               which Adam's m / sqrt(v) turns into +-lr steps of arbitrary sign on either side
        err = (sd[k].cpu().double() - v.detach()).abs().max().item()
        assert err < 3e-5, (k, err)                               # 3 steps of lr 1e-3 at the 1e-5-per-step budget


def test_fsasrec_with_dropout_runs_and_is_reproducible(tmp_path):
    from pixelrec_amd.model import FSASRec

    class DL:
        item_num = N

    outs = []
    for _ in range(2):
        torch.manual_seed(0)
        m = FSASRec(_config("fix", tmp_path, G["feats"], G["codes"], p_drop=0.2), DL()).cuda().train()
        loss = m((torch.from_numpy(G["items"]).cuda(), torch.from_numpy(G["masked_index"]).cuda()))
        loss.backward()
        outs.append((float(loss.detach()), _grads(m)["item_embedding.rec_fc.0.weight"].cpu()))
    assert np.isfinite(outs[0][0]) and outs[0][0] == outs[1][0] and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("fused", [True, False])
def test_trainer_runs_fsasrec_end_to_end(tmp_path, fused):
    """code/ViNet/sasrec_v.yaml-shaped run on TinyInter: Trainer.fit (hipGraph replay of the step included), evaluation through
    compute_item_all + predict / the fused top-k, a checkpoint with the reference's keys whose optimizer entry loads into the
    torch AdamW the reference Trainer builds (an EMPTY modal group + the rec group, trainer.py:93-96)."""
    from pixelrec_amd.config import Config
    from pixelrec_amd.data import bulid_dataloader, load_data
    from pixelrec_amd.optim import reference_rec_parameter_names
    from pixelrec_amd.parallel import DataParallel
    from pixelrec_amd.trainer import Trainer
    from pixelrec_amd.utils import get_model

    golden_dir = os.path.join(os.path.dirname(__file__), "golden")
    my, ov = tmp_path / "m.yaml", tmp_path / "o.yaml"
    my.write_text("model: FSASRec\nfreeze_model: True\nn_layers: 2\nn_heads: 2\nembedding_size: 32\ninner_size: 2\n"
                  "hidden_dropout_prob: 0.1\nattn_dropout_prob: 0.1\nhidden_act: 'gelu'\nlayer_norm_eps: 1e-12\n"
                  f"initializer_range: 0.02\nv_feat_path: {tmp_path}/feat.npy\n")
    ov.write_text(f"seed: 2020\nstate: INFO\nuse_modality: False\nreproducibility: True\ncheckpoint_dir: '{tmp_path}/saved'\n"
                  f"log_path: '{tmp_path}/log'\nshow_progress: False\nMAX_ITEM_LIST_LENGTH: 6\ndata_path: {golden_dir}/\n"
                  "dataset: TinyInter\nepochs: 3\ntrain_batch_size: 8\n"
                  "optim_args: {modal_lr: 0.0001, rec_lr: 0.003, modal_decay: 0.1, rec_decay: 0}\n"
                  "eval_batch_size: 16\ntopk: [5,10]\nmetrics: ['Recall', 'NDCG']\nvalid_metric: NDCG@10\n"
                  f"metric_decimal_place: 7\neval_step: 1\nstopping_step: 30\neval_fused_topk: {fused}\n")
    config = Config([str(my), str(ov)])
    config["device"] = torch.device("cuda", 0)
    dataload = load_data(config)
    train, valid, test = bulid_dataloader(config, dataload)
    np.save(str(tmp_path / "feat.npy"), np.random.default_rng(0).standard_normal((dataload.item_num, 24)).astype(np.float32))
    model = get_model(config["model"])(config, dataload)
    trainer = Trainer(config, DataParallel(model.to(config["device"])))
    trainer.fit(train, valid, saved=True)
    losses = [trainer.train_loss_dict[e] for e in sorted(trainer.train_loss_dict)]
    assert len(losses) == 3 and losses[-1] < losses[0]
    res = trainer.evaluate(test, load_best_model=True)
    assert set(res) == {"recall@5", "recall@10", "ndcg@5", "ndcg@10"}
    ck = torch.load(trainer.saved_model_file, map_location="cpu", weights_only=False)
    assert "item_embedding.rec_fc.0.weight" in ck["state_dict"] and "item_embedding.item_weights" not in ck["state_dict"]
    names = reference_rec_parameter_names(trainer.model.module)
    assert names[:2] == ["item_embedding.rec_fc.0.weight", "item_embedding.rec_fc.0.bias"] and names[2] == "position_embedding.weight"
    tparams = [torch.nn.Parameter(ck["state_dict"][k].clone()) for k in names]
    topt = torch.optim.AdamW([{"params": [], "lr": 1.0, "weight_decay": 0.5}, {"params": tparams, "lr": 1.0, "weight_decay": 0.5}])
    topt.load_state_dict(ck["optimizer"])
    assert topt.param_groups[1]["lr"] == 0.003 and topt.param_groups[1]["weight_decay"] == 0
    assert topt.state[tparams[0]]["exp_avg"].shape == tparams[0].shape
