"""Loss trajectory of N AdamW steps at batch size B with the sequence block on fp16 two-plane operands (PXR_SEQ_H2=1) and on the
six-product bf16x3 planes (0), same init / batches / dropout masks; prints both and the f32-mode trajectory for scale."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pixelrec_amd import ops  # noqa: E402
from pixelrec_amd.model import SASRec  # noqa: E402
from pixelrec_amd.optim import PxrAdamW  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
cfg = {"n_layers": 2, "n_heads": 4, "embedding_size": 512, "inner_size": 2, "hidden_dropout_prob": 0.1, "attn_dropout_prob": 0.1,
       "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02, "MAX_ITEM_LIST_LENGTH": 50, "seed": 2020}


class DL:
    item_num = 400001


def run(mode, gemm="bf16x3"):
    os.environ["PXR_SEQ_H2"] = mode
    prev = ops.set_gemm_mode(gemm)
    torch.manual_seed(0)
    m = SASRec(cfg, DL()).cuda()
    m.train()
    opt = PxrAdamW(m, lr=1e-4, weight_decay=0.1)
    g = torch.Generator().manual_seed(1)
    out = []
    for s in range(N):
        items = torch.randint(1, DL.item_num, (B, 2, 51), generator=g).cuda()
        mask = torch.ones(B, 50, dtype=torch.int64).cuda()
        opt.zero_grad()
        loss = m((items, mask))
        loss.backward()
        opt.step()
        out.append(float(loss))
    ops.raise_on_bad_indices("cuda")
    ops.set_gemm_mode(prev)
    return out


a, b, c = run("0"), run("1"), run("0", "f32")
for s in range(0, N, max(1, N // 10)):
    print("step %3d  bf16x3 %.6f  h2 %.6f  f32 %.6f   h2-bf16x3 %+.2e  f32-bf16x3 %+.2e" % (s, a[s], b[s], c[s], b[s] - a[s], c[s] - a[s]))
print("last      bf16x3 %.6f  h2 %.6f  f32 %.6f   h2-bf16x3 %+.2e  f32-bf16x3 %+.2e" % (a[-1], b[-1], c[-1], b[-1] - a[-1], c[-1] - a[-1]))
