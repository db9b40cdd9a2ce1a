"""GPU probe #1: gather + fp32 MFMA GEMM correctness vs torch-on-GPU and first timings. Scratch tool."""
import sys, time, json, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from pixelrec_amd import ops

dev = torch.device("cuda:0")
torch.manual_seed(0)
print("device", torch.cuda.get_device_name(0), torch.cuda.get_device_properties(0).total_memory / 2**30, "GiB")

def timeit(fn, iters=20, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3

res = {}
# ---------------- gather
for (N, D, n) in [(1000, 128, 333), (400001, 512, 6528), (400001, 512, 208896), (50000, 4096, 7777)]:
    table = torch.randn(N, D, device=dev)
    idx = torch.randint(0, N, (n,), device=dev)
    out = ops.embed_gather(table, idx)
    ref = table[idx]
    ok = torch.equal(out, ref)
    t = timeit(lambda: ops.embed_gather(table, idx))
    gbs = 2 * n * D * 4 / t / 1e9
    print(f"gather N={N} D={D} n={n}: exact={ok} {t*1e6:.1f} us {gbs:.0f} GB/s")
    res[f"gather_{N}_{D}_{n}"] = dict(ok=ok, us=t * 1e6, gbs=gbs)
    del table

# ---------------- HBM copy ceiling
x = torch.empty(1 << 28, device=dev); y = torch.empty_like(x)
t = timeit(lambda: y.copy_(x)); print(f"torch copy 1 GiB: {2*x.numel()*4/t/1e9:.0f} GB/s")
del x, y

# ---------------- GEMM
def check_gemm(M, N, K, kind, tile=0, split=0):
    if kind == "fwd":      # y = x W^T + b
        x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
        y = torch.empty(M, N, device=dev)
        f = lambda: ops.gemm(True, True, M, N, K, x, K, W, K, y, N, ops.EPI_BIAS, bias=b, use_ws=False, tile_hint=tile)
        f(); ref = (x.double() @ W.double().t() + b.double())
    elif kind == "dx":     # dx = dy W
        dy = torch.randn(M, K, device=dev); W = torch.randn(K, N, device=dev) * 0.05   # here GEMM (M,N,K): A[M,K] KC, B stored [K][N]
        y = torch.empty(M, N, device=dev)
        f = lambda: ops.gemm(True, False, M, N, K, dy, K, W, N, y, N, ops.EPI_NONE, use_ws=False, tile_hint=tile)
        f(); ref = dy.double() @ W.double()
    elif kind == "dw":     # C[M,N] = A^T B with A stored [K][M], B stored [K][N]
        A = torch.randn(K, M, device=dev); B = torch.randn(K, N, device=dev) * 0.05
        y = torch.empty(M, N, device=dev)
        f = lambda: ops.gemm(False, False, M, N, K, A, M, B, N, y, N, ops.EPI_NONE, use_ws=True, tile_hint=tile, split_hint=split)
        f(); ref = A.double().t() @ B.double()
    err = (y.double() - ref).abs().max().item()
    scale = ref.abs().max().item()
    t = timeit(f, iters=10)
    tf = 2.0 * M * N * K / t / 1e12
    print(f"gemm {kind:3s} M={M} N={N} K={K} tile={tile} split={split}: maxerr={err:.3e} (scale {scale:.2f}) {t*1e6:.1f} us {tf:.1f} TF")
    res[f"gemm_{kind}_{M}_{N}_{K}_{tile}_{split}"] = dict(err=err, us=t * 1e6, tf=tf)

for kind in ["fwd", "dx", "dw"]:
    check_gemm(80, 128, 128, kind)           # tiny, ragged M
    check_gemm(100, 132, 64, kind)           # ragged everything
check_gemm(76, 100, 52, "dw")                # ragged K
for tile in (64, 128):
    check_gemm(3200, 512, 512, "fwd", tile)
    check_gemm(3200, 1024, 512, "fwd", tile)
    check_gemm(3200, 512, 1024, "fwd", tile)
    check_gemm(3200, 512, 512, "dx", tile)
    check_gemm(512, 512, 3200, "dw", tile)
    check_gemm(512, 512, 3200, "dw", tile, 4)
    check_gemm(4096, 4096, 4096, "fwd", tile)
    check_gemm(1024, 400001, 512, "fwd", tile)
    check_gemm(102400, 512, 512, "fwd", tile)
# asymmetric transpose check with A = I
M = N = K = 128
x = torch.eye(128, device=dev); W = torch.arange(128 * 128, device=dev, dtype=torch.float32).reshape(128, 128)
y = torch.empty(128, 128, device=dev)
ops.gemm(True, True, M, N, K, x, K, W, K, y, N, ops.EPI_NONE, use_ws=False)
print("A=I check (y == W^T):", torch.equal(y, W.t().contiguous()))
t = timeit(lambda: torch.mm(torch.empty(4096, 4096, device=dev), torch.empty(4096, 4096, device=dev)), iters=5)
a = torch.randn(4096, 4096, device=dev); b = torch.randn(4096, 4096, device=dev)
t = timeit(lambda: torch.mm(a, b), iters=10); print(f"rocBLAS fp32 4096^3: {2*4096**3/t/1e12:.1f} TF")
a = torch.randn(1024, 512, device=dev); b = torch.randn(400001, 512, device=dev)
t = timeit(lambda: torch.mm(a, b.t()), iters=5); print(f"rocBLAS fp32 scoring 1024x400001x512: {2*1024*400001*512/t/1e12:.1f} TF  {t*1e3:.2f} ms")
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/probe1.json", "w"), indent=1)
