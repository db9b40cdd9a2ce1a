"""`decay_check_name` (reference code/REC/trainer/trainer.py:73-91): the two AdamW groups split by a name fragment.  The HIP
per-tensor optimizer (pixelrec_amd.optim.FragmentAdamW) against the oracle's AdamW with the same per-name (lr, weight_decay),
for fragments that cut through the flat parameter buffer, that pick the item table, and that match nothing."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

META = dict(n_items=300, D=64, L=10, H=2, inner=2, n_layers=2)
CFG = {"n_layers": 2, "n_heads": 2, "embedding_size": 64, "inner_size": 2, "hidden_dropout_prob": 0.0, "attn_dropout_prob": 0.0,
       "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02, "MAX_ITEM_LIST_LENGTH": 10, "seed": 2020}
IN, OUT = (3e-3, 0.02), (1e-3, 0.1)          # (lr, weight_decay) of the fragment group / of the rest


@pytest.mark.parametrize("fragment", ["LayerNorm", "item_embedding", "bias", "visual_encoder", "module."])
def test_fragment_groups_match_the_oracle(fragment):
    from oracle import sasrec_oracle as O
    from pixelrec_amd import synth
    from pixelrec_amd.model import SASRec
    from pixelrec_amd.optim import FragmentAdamW
    from pixelrec_amd.parallel import DataParallel

    class DL:
        item_num = META["n_items"]

    params = O.synth_params(META["n_items"], 64, 10, 2, 2, seed=4)
    model = SASRec(CFG, DL())
    model.load_state_dict(params, strict=True)
    dp = DataParallel(model.cuda().train())
    opt = FragmentAdamW(dp, fragment, *IN, *OUT)
    g0, g1 = opt.groups()
    assert all(n.startswith("module.") for n in g0 + g1) and len(g0) + len(g1) == len(params)
    assert len(g0) == sum(fragment in "module." + k for k in params)
    tr = O.OracleTrainer(params, {"n_layers": 2, "n_heads": 2, "layer_norm_eps": 1e-12},
                         group_of=lambda k: IN if fragment in "module." + k else OUT)
    rng = np.random.default_rng(1)
    z = synth.ZipfItems(META["n_items"], seed=1)
    for _ in range(3):
        it, mk = synth.train_batch(META["n_items"], 8, 10, rng, z)
        loss = dp((torch.from_numpy(it).cuda(), torch.from_numpy(mk).cuda()))
        loss.backward()
        opt.step()
        ref = tr.step(torch.from_numpy(it), torch.from_numpy(mk))
        assert abs(float(loss) - float(ref)) < 2e-5 * max(1.0, abs(float(ref)))
    tol = 2e-2 * max(IN[0] if g0 else 0.0, OUT[0] if g1 else 0.0)      # an AdamW step moves a weight by ~lr: 2 % of that (2e-5 at lr 1e-3)
    sd = model.state_dict()
    for k in params:
        assert (sd[k].cpu() - tr.p[k]).abs().max().item() < tol, k
    # the checkpoint has torch.optim.AdamW's shape: two groups, state numbered through them in order; a reload resumes it
    osd = opt.state_dict()
    assert [len(g["params"]) for g in osd["param_groups"]] == [len(g0), len(g1)]
    assert (osd["param_groups"][0]["lr"], osd["param_groups"][0]["weight_decay"]) == IN
    assert (osd["param_groups"][1]["lr"], osd["param_groups"][1]["weight_decay"]) == OUT
    names = g0 + g1
    for i, n in enumerate(names):
        assert (osd["state"][i]["exp_avg"].cpu() - tr.m[n[len("module."):]]).abs().max().item() < 1e-6, n
    opt2 = FragmentAdamW(dp, fragment, 1.0, 1.0, 1.0, 1.0)
    opt2.load_state_dict(osd)
    assert opt2.step_count == 3 and opt2.param_groups[0]["lr"] == IN[0] and opt2.param_groups[1]["weight_decay"] == OUT[1]
    it, mk = synth.train_batch(META["n_items"], 8, 10, rng, z)
    loss = dp((torch.from_numpy(it).cuda(), torch.from_numpy(mk).cuda()))
    loss.backward()
    opt2.step()
    tr.step(torch.from_numpy(it), torch.from_numpy(mk))
    sd = model.state_dict()
    for k in params:
        assert (sd[k].cpu() - tr.p[k]).abs().max().item() < tol, k
