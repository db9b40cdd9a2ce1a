"""Smaller contracts of the product path on the GPU: out-of-catalogue ids raise (like nn.Embedding in the reference),
optimizer state interchanges with torch.optim.AdamW's layout (reference trainer.py:153,186)."""
import numpy as np
import pytest
import torch

from oracle import sasrec_oracle as O

pytestmark = pytest.mark.gpu

N, D, L, H, B = 300, 64, 8, 2, 5
CFG = {"n_layers": 2, "n_heads": H, "embedding_size": D, "inner_size": 2, "hidden_dropout_prob": 0.0,
       "attn_dropout_prob": 0.0, "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02,
       "MAX_ITEM_LIST_LENGTH": L, "seed": 2020}


class DL:
    item_num = N


def _model(params):
    from pixelrec_amd.model import SASRec

    m = SASRec(CFG, DL())
    m.load_state_dict(params, strict=True)
    return m.cuda().train()


def _batches(n, seed=1):
    from pixelrec_amd import synth

    rng = np.random.default_rng(seed)
    z = synth.ZipfItems(N, seed=seed)
    return [tuple(torch.from_numpy(a) for a in synth.train_batch(N, B, L, rng, z)) for _ in range(n)]


def test_out_of_range_item_ids_raise_index_error():
    """sasrec.py:68,101: nn.Embedding raises on an id outside [0, item_num).  Here the gather kernels flag it in the
    device status word (and clamp, so nothing faults) and the host raises at its next check."""
    from pixelrec_amd import ops

    table = torch.randn(N, D, device="cuda")
    ops.raise_on_bad_indices()                                        # clean slate
    ok = ops.embed_gather(table, torch.tensor([0, 1, N - 1], device="cuda"))
    ops.raise_on_bad_indices()                                        # in-range ids: nothing flagged
    assert torch.equal(ok, table[[0, 1, N - 1]])
    ops.embed_gather(table, torch.tensor([3, N, 7], device="cuda"))
    with pytest.raises(IndexError):
        ops.raise_on_bad_indices()
    ops.raise_on_bad_indices()                                        # the flag was consumed
    ops.embed_gather(table, torch.tensor([-1], device="cuda"))
    with pytest.raises(IndexError):
        ops.raise_on_bad_indices()
    # through the model: predict() checks after scoring (the fused gather + LayerNorm kernel flags the id)
    m = _model(O.synth_params(N, D, L, 2, 2, seed=3)).eval()
    seq = torch.randint(1, N, (4, L), device="cuda")
    m.predict(seq, m.compute_item_all())
    seq[2, -1] = N + 5
    with pytest.raises(IndexError):
        m.predict(seq, m.compute_item_all())


def test_optimizer_state_interchanges_with_torch_adamw_layout():
    """PxrAdamW.state_dict(layout="torch") is what torch.optim.AdamW.state_dict() holds for the reference model after the
    same steps (per-parameter step / exp_avg / exp_avg_sq in the reference's parameter order; values against the CPU
    oracle's AdamW moments, itself pinned on 4 torch.optim.AdamW steps of the reference), and a torch-layout state
    resumes bit-exactly."""
    from pixelrec_amd.optim import PxrAdamW, reference_rec_parameter_names

    params = O.synth_params(N, D, L, 2, 2, seed=5, perturb=True)
    batches = _batches(5)
    m = _model(params)
    opt = PxrAdamW(m, lr=1e-3, weight_decay=0.1)
    tr = O.OracleTrainer(params, {"n_layers": 2, "n_heads": H, "layer_norm_eps": 1e-12}, lr=1e-3, weight_decay=0.1)
    for it, mk in batches[:3]:
        m((it.cuda(), mk.cuda())).backward()
        opt.step()
        tr.step(it, mk)
    sd = opt.state_dict(layout="torch")
    names = reference_rec_parameter_names(m)
    assert names[0] == "item_embedding.weight" and names[1] == "position_embedding.weight" and names[-1] == "LayerNorm.bias"
    assert set(sd) == {"state", "param_groups"} and sd["param_groups"][0]["params"] == list(range(len(names)))
    assert sd["param_groups"][0]["lr"] == 1e-3 and sd["param_groups"][0]["weight_decay"] == 0.1
    assert names == list(tr.p.keys())       # the oracle's parameters are in the reference's registration order
    for i, name in enumerate(names):        # the oracle's moments are torch.optim.AdamW's exp_avg / exp_avg_sq
        st = sd["state"][i]
        assert float(st["step"]) == float(tr.t) == 3.0
        assert st["exp_avg"].shape == tr.p[name].shape
        for k, rs in (("exp_avg", tr.m[name]), ("exp_avg_sq", tr.v[name])):
            err = (st[k].cpu() - rs).abs().max().item()
            assert err <= 1e-9 + 2e-4 * rs.abs().max().item(), (name, k, err)
    # resume from the torch layout (as a reference checkpoint would provide it) == uninterrupted run, bit for bit
    cont = {k: v.detach().clone() for k, v in m.state_dict().items()}
    m2 = _model(cont)
    opt2 = PxrAdamW(m2, lr=5e-4, weight_decay=0.0)                   # overwritten by the loaded param_groups
    opt2.load_state_dict({"state": {i: {k: (v.cpu() if torch.is_tensor(v) else v) for k, v in s.items()}
                                    for i, s in sd["state"].items()}, "param_groups": sd["param_groups"]})
    assert opt2.step_count == 3 and opt2.param_groups[0]["lr"] == 1e-3
    for it, mk in batches[3:]:
        m((it.cuda(), mk.cuda())).backward(); opt.step()
        m2((it.cuda(), mk.cuda())).backward(); opt2.step()
    a, b = m.state_dict(), m2.state_dict()
    for k in a:
        assert torch.equal(a[k], b[k]), k
