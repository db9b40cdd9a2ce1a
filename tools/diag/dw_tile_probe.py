import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pixelrec_amd import ops
def timeit(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
K = 3200
for M, N in ((2048, 2048), (1536, 512), (4096, 1024)):
    A = torch.randn(K, M, device="cuda"); B = torch.randn(K, N, device="cuda"); C = torch.empty(M, N, device="cuda")
    r = {}
    for rnd in range(3):
        for name, th in (("b3_64", 9064), ("b3_1281", 91281), ("f32_64", 64)):
            r.setdefault(name, []).append(timeit(lambda: ops.gemm(False, False, M, N, K, A, M, B, N, C, N, ops.EPI_NONE, use_ws=False, tile_hint=th, split_hint=1)))
    fl = 2.0 * M * N * K
    print(M, N, {k: (round(sorted(v)[1], 1), round(fl / sorted(v)[1] / 1e6, 1)) for k, v in r.items()})
