import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pixelrec_amd import ops
def t_us(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (M, N, K) in ((17600, 3072, 768), (17600, 768, 3072), (102400, 1536, 512), (1024, 400001, 512)):
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.02; b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    line = f"M={M} N={N} K={K}:"
    for tile in (128, 1281, 1282, 12861):
        t = t_us(lambda: ops.gemm(True, True, M, N, K, x, K, W, K, y, N, ops.EPI_BIAS, bias=b, use_ws=False, tile_hint=tile))
        line += f"  t{tile} {2.0*M*N*K/t/1e6:6.1f} TF"
    print(line)
