"""Single-GPU timing of the data-parallel merge step at world sizes 2/4/8 (synthetic Zipf rank-local lists of the
north-star shape): rank-ordered merge (pxr_merge_sorted_rows_f32) vs re-sorting the concatenation
(pxr_embed_grad_rows_f32)."""
import numpy as np
import torch

from pixelrec_amd import ops, synth
from pixelrec_amd.parallel import PAD_ID

N, D, B, L = 400001, 512, 64, 50
cap = B * (2 * L + 1)
z = synth.ZipfItems(N, seed=1)


def t_us(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


for W in (1, 2, 4, 8):
    idx_all = torch.full((W, cap), PAD_ID, dtype=torch.int64)
    nu = []
    for r in range(W):
        rng = np.random.default_rng(10 + r)
        it, _ = synth.train_batch(N, B, L, rng, z)
        u = np.unique(it)
        u = u[u > 0]
        idx_all[r, :len(u)] = torch.from_numpy(u)
        nu.append(len(u))
    idx_all = idx_all.reshape(-1).cuda()
    rows_all = torch.randn(W * cap, D, device="cuda")
    out_m = ops.SparseRows(W * cap, D, "cuda")
    out_s = ops.SparseRows(W * cap, D, "cuda")
    tm = t_us(lambda: ops.merge_sorted_rows(idx_all, rows_all, W, N, 1.0, out=out_m))
    ts = t_us(lambda: ops.embed_grad_rows(idx_all, rows_all, N, 1.0, out=out_s))
    a, b = out_m.to_dense(N), out_s.to_dense(N)
    print(f"W={W}: unique/rank ~{int(np.mean(nu))} of cap {cap}; union {out_s.count()}; merge {tm:.1f} us, "
          f"re-sort {ts:.1f} us; max |diff| {float((a - b).abs().max()):.2e}")
