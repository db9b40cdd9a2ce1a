"""Step time of the sibling backbones at their shipped shapes (IDNet/gru4rec.yaml: emb 2048, hidden 1x, 1 layer; overall/ID.yaml:
B = 64, L = 10) on the synthetic 400 001-item catalogue -- hipGraph replay of the whole training step, like bench.py.
usage (GPU box): python tools/sibling_bench.py"""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

from pixelrec_amd import synth
from pixelrec_amd.graph import GraphedTrainStep
from pixelrec_amd.model import GRU4Rec
from pixelrec_amd.optim import PxrAdamW
from pixelrec_amd.parallel import DataParallel


def main():
    N, B, L = 400001, 64, 10
    out = []
    for E, mult in ((2048, 1), (512, 1)):
        class DL:
            item_num = N

        torch.manual_seed(0)
        m = GRU4Rec({"embedding_size": E, "hidden_size": mult, "num_layers": 1, "dropout_prob": 0.0, "MAX_ITEM_LIST_LENGTH": L,
                     "seed": 2020}, DL()).cuda().train()
        dp = DataParallel(m)
        opt = PxrAdamW(m, lr=1e-4, weight_decay=0.1)
        rng = np.random.default_rng(1)
        zipf = synth.ZipfItems(N, seed=2020)
        batches = [tuple(torch.from_numpy(a).cuda() for a in synth.train_batch(N, B, L, rng, zipf)) for _ in range(60)]
        g = GraphedTrainStep(dp, opt, *batches[0])
        for b in batches[:20]:
            g(*b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for b in batches[20:]:
            g(*b)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 40
        H = E * mult
        flops = 2.0 * B * L * (3 * H * E + 3 * H * H + H * E) * 3        # fwd + dX + dW of the three linear maps
        r = {"model": "GRU4Rec", "embedding_size": E, "hidden": H, "batch": B, "seq_len": L, "ms_per_step": ms,
             "sequences_per_s": B / ms * 1e3, "algorithmic_gflop_per_step": flops / 1e9, "final_loss": float(g.loss)}
        print(json.dumps(r)); out.append(r)
        del m, dp, opt, g
        torch.cuda.empty_cache()
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/sibling_bench.json", "w"), indent=1)


if __name__ == "__main__":
    main()
