"""The product's planes GEMM entry point at the ViT-tower / scoring shapes: lockstep tile (gemm_p3.cuh) vs the ping-pong tiles
(gemm_p4.cuh), per epilogue.  Prints us, TFLOP/s with the six bf16 products counted, and the share of the 2.5 PFLOP/s bf16 pipe."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pixelrec_amd import ops


def t_us(fn, n=8):
    for _ in range(2): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3


TILES = ((825612820, "p3 256x128"), (425612833, "p4 256x128 a3"), (425612832, "p4 256x128 a2"), (425625631, "p4 256x256 a1"))
SHAPES = (("qkv", 69344, 2304, 768), ("out", 69344, 768, 768), ("fc1", 69344, 3072, 768), ("fc2", 69344, 768, 3072))
only = sys.argv[1:] or None
for name, M, N, K in SHAPES:
    if only and name not in only: continue
    x = torch.randn(M, K, device="cuda"); W = torch.randn(N, K, device="cuda") * 0.03; b = torch.randn(N, device="cuda")
    res = torch.randn(M, N, device="cuda")
    xp, Wp = ops.split_planes(x), ops.split_planes(W)
    y = torch.empty(M, N, device="cuda"); yp = ops.Planes.alloc(M, N, "cuda")
    cases = (("bias -> fp32", ops.EPI_BIAS, None, y, None), ("bias+residual -> fp32", ops.EPI_BIAS_ADD, res, y, None),
             ("bias+qgelu -> planes", ops.EPI_BIAS_QGELU, None, None, yp), ("bias -> fp32 + planes", ops.EPI_BIAS, None, y, yp))
    for cname, epi, aux, C, Cp in cases:
        line = f"{name} M={M} N={N} K={K}  {cname:24s}"
        for code, tname in TILES:
            t = t_us(lambda: ops.gemm_planes(xp, Wp, C, epi, bias=b, aux=aux, tile_hint=code, Cp=Cp))
            tf = 12.0 * M * N * K / t / 1e6
            line += f" | {tname} {t:7.1f} us {tf / 2500:5.3f}"
        print(line, flush=True)
