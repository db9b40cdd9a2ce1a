"""Full-sort eval: fused scoring+mask+top-k vs the literal predict -> mask -> torch.topk sequence (scratch tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pixelrec_amd import ops, synth
dev = torch.device("cuda:0")
def timeit(fn, iters=10, warm=2):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters
N, D, B = 400001, 512, 1024
table = torch.randn(N, D, device=dev) * 0.02
users = torch.randn(B, D, device=dev)
rng = np.random.default_rng(0)
_, hu, hi, _ = synth.eval_batch(N, B, 50, rng, synth.ZipfItems(N))
hu, hi = torch.from_numpy(hu), torch.from_numpy(hi)
ptr, items = ops.history_csr(hu, hi, B, dev)
hud, hid = hu.to(dev), hi.to(dev)
scores = torch.empty(B, N, device=dev)
def literal():
    ops.gemm(True, True, B, N, D, users, D, table, D, scores, N, ops.EPI_NONE, use_ws=False)
    scores[:, 0] = -np.inf
    scores[(hud, hid)] = -np.inf
    return torch.topk(scores, 10, dim=-1)
def fused():
    return ops.score_topk(users, D, B, table, 10, ptr, items)
v1, i1 = literal(); i2, v2 = fused()
print("identical top-10:", torch.equal(i1, i2), "max |dval|", (v1 - v2).abs().max().item())
tl, tf = timeit(literal), timeit(fused)
fl = 2.0 * B * N * D
print(f"literal (gemm + masks + torch.topk): {tl:.2f} ms   fused: {tf:.2f} ms   speed-up {tl/tf:.2f}x   fused effective {fl/tf/1e9:.1f} TFLOP/s ({fl/tf/1e9/157.3*100:.0f}% of fp32-MFMA peak)")
