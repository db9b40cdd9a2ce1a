"""Collector / Recall / NDCG / Evaluator against the reference's outputs stored in the goldens."""
import numpy as np
import pytest
import torch

from pixelrec_amd.evaluator import Collector, Evaluator
from tests.golden_util import CASES, load_case

CFG = {"metrics": ["Recall", "NDCG"], "topk": [5, 10], "metric_decimal_place": 7}


@pytest.mark.parametrize("case", CASES)
def test_collector_and_metrics_match_reference(case):
    meta, z = load_case(case)
    Be = z["eval.item_seq"].shape[0]
    n = meta["n_items"]
    # rebuild a masked score matrix whose top-10 equals the reference's (values from the golden top-k)
    scores = torch.full((Be, n), -np.inf)
    scores.scatter_(1, torch.from_numpy(z["eval.topk_idx"]), torch.from_numpy(z["eval.topk_val"]))
    coll, ev = Collector(CFG), Evaluator(CFG)
    coll.eval_batch_collect(scores, torch.arange(Be), torch.from_numpy(z["eval.positive_i_planted"]))
    st = coll.get_data_struct()
    assert np.array_equal(st.get("rec.topk").numpy(), z["eval.rec_topk"])
    res = ev.evaluate(st)
    assert list(res.keys()) == [str(x) for x in z["eval.metric_names"]]
    assert np.allclose(list(res.values()), z["eval.metric_sums"], atol=1e-12)
    # the fused-kernel entry gives the same rec.topk from the top-k ids alone
    coll.eval_topk_collect(torch.from_numpy(z["eval.topk_idx"]), torch.from_numpy(z["eval.positive_i_planted"]))
    assert np.array_equal(coll.get_data_struct().get("rec.topk").numpy(), z["eval.rec_topk"])
