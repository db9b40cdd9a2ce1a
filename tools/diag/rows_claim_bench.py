"""What bounds the claim-mode catch-up of the lazy table AdamW (csrc/adamw.hip, pxr_adamw_rows_ids2d_f32) at the bench shape: the ids
of a Zipf batch (duplicates: every occurrence raises last[row] with an atomic) against the same number of distinct ids, rows that
need a replay against rows that are current.  python tools/diag/rows_claim_bench.py"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from pixelrec_amd import ops, synth  # noqa: E402

N, D, B, L, T = 400_001, 512, 64, 50, 700
dev = "cuda"
g = torch.Generator().manual_seed(0)
p = (torch.randn(N, D, generator=g) * 0.02).to(dev)
m = (torch.randn(N, D, generator=g) * 1e-4).to(dev)
v = (torch.rand(N, D, generator=g) * 1e-8).to(dev)
hyper = torch.zeros(T + 8, 4, device=dev)
cumlog = torch.zeros(T + 8, dtype=torch.float64, device=dev)
for t in range(1, T + 1):
    ops.adamw_hyper_append(hyper, cumlog, t, 1e-4, 0.9, 0.999, 1e-8, 0.1)
rng = np.random.default_rng(1)
zipf = synth.ZipfItems(N, seed=2020)
items_z = torch.from_numpy(synth.train_batch(N, B, L, rng, zipf, full_frac=1.0)[0]).to(dev)
items_u = items_z.clone()
items_u[:, 0, :] = torch.from_numpy(rng.permutation(N - 1)[:B * (L + 1)] + 1).view(B, L + 1).to(dev)
print("distinct input ids: zipf", items_z[:, 0, :L].unique().numel(), "distinct", items_u[:, 0, :L].unique().numel(), "of", B * L,
      " most frequent id occurs", int(torch.bincount(items_z[:, 0, :L].flatten()).max()), "times")
last0 = {"gap50": torch.full((N,), T - 50, dtype=torch.int32, device=dev), "current": torch.full((N,), T, dtype=torch.int32, device=dev),
         "gap1": torch.full((N,), T - 1, dtype=torch.int32, device=dev)}
for ids_name, items in (("zipf", items_z), ("distinct", items_u)):
    for st_name, l0 in last0.items():
        last = l0.clone()
        ts = []
        for it in range(12):
            last.copy_(l0)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            ops.adamw_rows_ids2d(p, m, v, last, hyper, cumlog, T, 0.9, 0.999, 1e-8, items, B, L, 2 * (L + 1))
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
        ts = sorted(ts[2:])
        print(f"{ids_name:9s} {st_name:8s} median {ts[len(ts) // 2]:6.1f} us   min {ts[0]:6.1f}")
