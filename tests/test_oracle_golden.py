"""Pins the CPU oracle (oracle/sasrec_oracle.py) against golden vectors produced by the REFERENCE itself
(oracle/make_golden.py; the reference has no tests of its own).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import sasrec_oracle as O
from tests.golden_util import CASES, compare, load_case, oracle_cfg


def _params(meta):
    return O.synth_params(meta["n_items"], meta["D"], meta["L"], meta["n_layers"], meta["inner"], seed=meta["seed"])


@pytest.mark.parametrize("case", CASES)
def test_forward_activations_and_loss(case):
    meta, z = load_case(case)
    p = _params(meta)
    trace = {}
    loss = O.forward_loss(p, torch.from_numpy(z["items"]), torch.from_numpy(z["masked_index"]), oracle_cfg(meta), None, trace)
    assert abs(float(loss) - float(z["loss"])) <= 2e-5 * max(1.0, abs(float(z["loss"])))
    compare(z, "act.input_emb", trace["input_emb"], 2e-5)
    for i in range(meta["n_layers"]):
        compare(z, f"act.layer{i}.attn_out", trace[f"layer{i}.attn_out"], 3e-5)
        compare(z, f"act.layer{i}.ffn_out", trace[f"layer{i}.ffn_out"], 3e-5)
    compare(z, "pos_score", trace["pos_score"], 2e-5)
    compare(z, "neg_score", trace["neg_score"], 2e-5)


@pytest.mark.parametrize("case", CASES)
def test_gradients(case):
    meta, z = load_case(case)
    p = _params(meta)
    _, g = O.loss_and_grads(p, torch.from_numpy(z["items"]), torch.from_numpy(z["masked_index"]), oracle_cfg(meta))
    for k, v in g.items():
        if k == "item_embedding.weight":
            rows = z["grad.item_embedding.rows"]
            nz = torch.nonzero(v.abs().sum(1) > 0).squeeze(1).numpy()
            assert np.array_equal(nz, rows)
            assert float(v[0].abs().max()) == 0.0
            compare(z, "grad.item_embedding.vals", v[torch.from_numpy(rows)], 2e-6, 1e-4)
            assert np.allclose(v.double().sum(1).numpy(), z["grad.item_embedding.rowsum"], atol=2e-6)
        else:
            compare(z, "grad." + k, v, 3e-6, 1e-4)


@pytest.mark.parametrize("case", CASES)
def test_adamw_four_steps(case):
    meta, z = load_case(case)
    tr = O.OracleTrainer(_params(meta), oracle_cfg(meta), lr=1e-4, weight_decay=0.1)
    rows = torch.from_numpy(z["adamw.watch_rows"])
    for s in range(4):
        loss = tr.step(torch.from_numpy(z["adamw.items"][s]), torch.from_numpy(z["adamw.masks"][s]))
        assert abs(float(loss) - float(z[f"adamw.loss{s}"])) <= 3e-5 * max(1.0, abs(float(loss)))
        # post-step parameters: +-1e-5 (BASELINE.md parity gate); AdamW's g/|g| is sign-sensitive only where
        # |g| is at rounding-noise level, which these watched tensors do not hit in the goldens.
        compare(z, f"adamw.step{s}.table_rows", tr.p["item_embedding.weight"][rows], 1e-5)
        assert abs(float(tr.p["item_embedding.weight"].double().sum()) - float(z[f"adamw.step{s}.table_sum"])) < 1e-3
        for k in ("position_embedding.weight", "LayerNorm.weight", "LayerNorm.bias",
                  "trm_encoder.layer.0.multi_head_attention.query.weight",
                  "trm_encoder.layer.1.feed_forward.dense_2.weight",
                  "trm_encoder.layer.1.feed_forward.dense_1.bias",
                  "trm_encoder.layer.0.feed_forward.LayerNorm.weight"):
            compare(z, f"adamw.step{s}." + k, tr.p[k], 1e-5)


@pytest.mark.parametrize("case", CASES)
def test_predict_topk_metrics(case):
    meta, z = load_case(case)
    p = _params(meta)
    scores = O.predict(p, torch.from_numpy(z["eval.item_seq"]), p["item_embedding.weight"], oracle_cfg(meta))
    cs = int(z["eval.scores_colstride"])
    assert np.abs(scores[:, ::cs].numpy() - z["eval.scores"]).max() <= 1e-4   # logits +-1e-4 (BASELINE.json)
    masked = O.full_sort_scores(scores, torch.from_numpy(z["eval.history_u"]), torch.from_numpy(z["eval.history_i"]))
    rec, idx = O.topk_hits(masked, torch.arange(scores.shape[0]), torch.from_numpy(z["eval.positive_i_planted"]), 10)
    assert np.array_equal(idx.numpy(), z["eval.topk_idx"])
    assert np.array_equal(rec.numpy(), z["eval.rec_topk"])
    res = O.recall_ndcg(rec.numpy(), [5, 10])
    for name, val in zip(z["eval.metric_names"], z["eval.metric_sums"]):
        assert abs(res[str(name)] - float(val)) < 1e-9, name


@pytest.mark.parametrize("act", ["relu", "swish", "tanh", "sigmoid"])
def test_oracle_hidden_act_matches_reference_golden(act):
    """The reference's other FeedForward activations (layers.py:642-649): tests/golden/sasrec_act.npz holds the reference
    SASRec's loss / gradients / scores per activation (oracle/make_golden_act.py)."""
    import os

    from tests.golden_util import GOLDEN_DIR, META_KEYS

    z = np.load(os.path.join(GOLDEN_DIR, "sasrec_act.npz"))
    meta = dict(zip(META_KEYS, [int(x) for x in z["meta"]]))
    p = O.synth_params(meta["n_items"], meta["D"], meta["L"], meta["n_layers"], meta["inner"], seed=meta["seed"])
    cfg = {"n_layers": meta["n_layers"], "n_heads": meta["H"], "layer_norm_eps": 1e-12, "hidden_act": act}
    loss, g = O.loss_and_grads(p, torch.from_numpy(z["items"]), torch.from_numpy(z["masked_index"]), cfg)
    assert abs(float(loss) - float(z[f"{act}.loss"])) < 1e-6
    for k in z.files:
        if k.startswith(f"{act}.grad."):
            name = k[len(act) + 6:]
            assert np.abs(g[name].numpy() - z[k]).max() < 2e-7 + 2e-6 * np.abs(z[k]).max(), name
    seq = torch.from_numpy(z["eval.item_seq"])
    scores = O.predict(p, seq, p["item_embedding.weight"], cfg)
    assert np.abs(scores.numpy() - z[f"{act}.scores"]).max() < 1e-6
