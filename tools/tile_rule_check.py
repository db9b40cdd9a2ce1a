"""Does the 16-wave 128x128 tile beat 64x64 whenever the problem has 160..256 tiles of 128x128 (one workgroup per CU)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixelrec_amd import ops
def t_us(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3
for (M, N, K) in ((3200, 1024, 512), (2400, 1536, 512), (3200, 1024, 1024), (4096, 1024, 512), (2048, 2048, 512), (3200, 896, 512),
                  (1600, 2048, 1024), (2560, 1280, 512), (4800, 1024, 512), (3200, 1280, 512), (3200, 768, 512)):
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.02; b = torch.randn(N, device="cuda")
    y = torch.empty(M, N, device="cuda")
    t128 = ((M + 127) // 128) * ((N + 127) // 128)
    r = {}
    for tile in (64, 1281):
        r[tile] = t_us(lambda: ops.gemm(True, True, M, N, K, x, K, W, K, y, N, ops.EPI_BIAS, bias=b, use_ws=False, tile_hint=tile))
    print(f"M={M} N={N} K={K} t128={t128:4d}: t64 {r[64]:6.1f} us  t1281 {r[1281]:6.1f} us  -> {'1281' if r[1281] < r[64] else '64'} ({(r[64]/r[1281]-1)*100:+.0f}%)")
