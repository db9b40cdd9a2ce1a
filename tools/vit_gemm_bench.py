"""fp32 Linear at the CLIP ViT-B shapes: torch (hipBLASLt / rocBLAS) vs the build's fp32-MFMA GEMM (scratch tool)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from pixelrec_amd import ops

def t_us(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / n * 1e3

for tokens in (352 * 50, 352 * 197):
    for n, k in ((2304, 768), (768, 768), (3072, 768), (768, 3072)):
        x = torch.randn(tokens, k, device="cuda"); W = torch.randn(n, k, device="cuda") * 0.02; b = torch.randn(n, device="cuda")
        a = t_us(lambda: F.linear(x, W, b)); m = t_us(lambda: ops.linear_fwd(x, W, b))
        fl = 2.0 * tokens * n * k
        print(f"M={tokens:6d} N={n:5d} K={k:5d}  torch {a:8.1f} us {fl / a / 1e6:6.1f} TF | pxr {m:8.1f} us {fl / m / 1e6:6.1f} TF")
