"""Sequence block at a given batch size: loss and gradients with the GEMMs on fp16 two-plane operands (PXR_SEQ_H2=1) against the
six-product bf16x3 planes (PXR_SEQ_H2=0) on the same weights, batch and dropout masks."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pixelrec_amd.model import SASRec  # noqa: E402
from pixelrec_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
pd = float(sys.argv[2]) if len(sys.argv) > 2 else 0.1
cfg = {"n_layers": 2, "n_heads": 4, "embedding_size": 512, "inner_size": 2, "hidden_dropout_prob": pd, "attn_dropout_prob": pd,
       "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02, "MAX_ITEM_LIST_LENGTH": 50, "seed": 2020}


class DL:
    item_num = 400001


torch.manual_seed(0)
m = SASRec(cfg, DL()).cuda()
m.train()
g = torch.Generator().manual_seed(1)
items = torch.randint(1, DL.item_num, (B, 2, 51), generator=g).cuda()
mask = torch.ones(B, 50, dtype=torch.int64).cuda()
res = {}
for mode in ("0", "1", "0"):
    os.environ["PXR_SEQ_H2"] = mode
    m._step_counter = 0
    if m._drop_dev is not None:
        m._drop_dev.fill_(0)
    loss = m((items, mask))
    loss.backward()
    flat, gflat = m.flat_parameters()
    res.setdefault(mode, []).append((float(loss), gflat.clone(), m.sparse_table_grad.rows[: int(m.sparse_table_grad.count())].clone()
                                     if hasattr(m.sparse_table_grad, "rows") else None))
    ops.raise_on_bad_indices("cuda")
l0, g0, r0 = res["0"][0]
l0b, g0b, r0b = res["0"][1]
l1, g1, r1 = res["1"][0]
print("loss  bf16x3 %.9f  (rerun %.9f)   h2 %.9f" % (l0, l0b, l1))
print("flat grad: max |bf16x3| %.3e   rerun diff %.3e   h2 diff %.3e   (rel to max %.3e)" %
      (g0.abs().max(), (g0 - g0b).abs().max(), (g0 - g1).abs().max(), (g0 - g1).abs().max() / g0.abs().max()))
if r0 is not None:
    print("table grad rows: max %.3e  rerun diff %.3e  h2 diff %.3e" % (r0.abs().max(), (r0 - r0b).abs().max(), (r0 - r1).abs().max()))
