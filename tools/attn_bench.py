import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixelrec_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=50, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e3
H, d = 4, 128
for B, L in [(64, 50), (64, 64), (128, 50), (256, 50), (512, 50), (2048, 50)]:
    D = H * d
    qkv = torch.randn(B, L, 3 * D, device=dev)
    km = torch.ones(B, L, dtype=torch.int64, device=dev)
    ctx, probs = ops.attn_fwd(qkv, km, L, B, H, L, d)
    dctx = torch.randn_like(ctx)
    tf = timeit(lambda: ops.attn_fwd(qkv, km, L, B, H, L, d))
    tb = timeit(lambda: ops.attn_bwd(dctx, qkv, probs, B, H, L, d))
    print(f"B={B:5d} L={L}: fwd {tf:7.1f} us  bwd {tb:7.1f} us   per-round(256 WGs) fwd {tf/max(1,B*H/256):6.1f} bwd {tb/max(1,B*H/256):6.1f}")
