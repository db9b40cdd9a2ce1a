"""GEMM tile sweep at the training-step shapes (scratch tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixelrec_amd import ops
dev = torch.device("cuda:0")
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
M = int(os.environ.get("M", 3200))
shapes = [("fwd", M, 1536, 512), ("fwd", M, 512, 512), ("fwd", M, 1024, 512), ("fwd", M, 512, 1024),
          ("dx", M, 512, 1536), ("dx", M, 512, 512), ("dx", M, 1024, 512), ("dx", M, 512, 1024),
          ("dw", 1536, 512, M), ("dw", 512, 512, M), ("dw", 1024, 512, M), ("dw", 512, 1024, M)]
tot = {}
for kind, m, n, k in shapes:
    line = f"{kind:3s} M={m:6d} N={n:5d} K={k:6d} ideal={2*m*n*k/157.3e12*1e6:6.1f}us |"
    for tile in [int(t) for t in os.environ.get("TILES", "64,642,128").split(",")]:
        if kind == "fwd":
            x = torch.randn(m, k, device=dev); W = torch.randn(n, k, device=dev); b = torch.randn(n, device=dev); y = torch.empty(m, n, device=dev)
            f = lambda: ops.gemm(True, True, m, n, k, x, k, W, k, y, n, ops.EPI_BIAS, bias=b, use_ws=False, tile_hint=tile)
        elif kind == "dx":
            dy = torch.randn(m, k, device=dev); W = torch.randn(k, n, device=dev); y = torch.empty(m, n, device=dev)
            f = lambda: ops.gemm(True, False, m, n, k, dy, k, W, n, y, n, ops.EPI_NONE, use_ws=False, tile_hint=tile)
        else:
            A = torch.randn(k, m, device=dev); Bm = torch.randn(k, n, device=dev); y = torch.empty(m, n, device=dev)
            f = lambda: ops.gemm(False, False, m, n, k, A, m, Bm, n, y, n, ops.EPI_NONE, use_ws=True, tile_hint=tile)
        t = timeit(f)
        tot[tile] = tot.get(tile, 0) + t
        line += f" t{tile}: {t*1e6:6.1f}us {2*m*n*k/t/1e12:5.1f}TF |"
    print(line)
print("sum per tile (us):", {k: round(v * 1e6, 1) for k, v in tot.items()})
