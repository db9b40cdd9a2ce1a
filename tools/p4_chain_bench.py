"""fc1 -> fc2 of a ViT-B/16 block back to back, as vit_native issues them (fc1 writes the planes fc2 reads; fc2 adds the residual):
per-kernel times inside the chain vs each GEMM alone, to see what the producer's dirty lines / the chip's sustained clock cost."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from pixelrec_amd import ops

M, H, I = 69344, 768, 3072
x = torch.randn(M, H, device="cuda"); W1 = torch.randn(I, H, device="cuda") * 0.03; W2 = torch.randn(H, I, device="cuda") * 0.03
b1 = torch.randn(I, device="cuda"); b2 = torch.randn(H, device="cuda"); res = torch.randn(M, H, device="cuda")
xp, W1p, W2p = ops.split_planes(x), ops.split_planes(W1), ops.split_planes(W2)
fp = ops.Planes.alloc(M, I, "cuda"); y = torch.empty(M, H, device="cuda")
ev = lambda: torch.cuda.Event(enable_timing=True)
def fc1(): ops.gemm_planes(xp, W1p, None, ops.EPI_BIAS_QGELU, bias=b1, Cp=fp)
def fc2(): ops.gemm_planes(fp, W2p, y, ops.EPI_BIAS_ADD, bias=b2, aux=res)
for name, seq in (("fc1 alone x8", [fc1] * 8), ("fc2 alone x8", [fc2] * 8), ("chain (fc1, fc2) x12", [fc1, fc2] * 12), ("chain x40 (sustained)", [fc1, fc2] * 40)):
    for f in seq[:2]: f()
    torch.cuda.synchronize()
    marks = [ev() for _ in range(len(seq) + 1)]
    marks[0].record()
    for i, f in enumerate(seq):
        f(); marks[i + 1].record()
    torch.cuda.synchronize()
    t = [marks[i].elapsed_time(marks[i + 1]) * 1e3 for i in range(len(seq))]
    a = [t[i] for i in range(len(seq)) if seq[i] is fc1]; b = [t[i] for i in range(len(seq)) if seq[i] is fc2]
    half = lambda v: sum(v[len(v) // 2:]) / max(len(v) - len(v) // 2, 1)
    print(f"{name:24s} fc1 {sum(a) / max(len(a), 1):7.1f} us (2nd half {half(a):7.1f})   fc2 {sum(b) / max(len(b), 1):7.1f} us (2nd half {half(b):7.1f})", flush=True)
