"""fp32-MFMA GEMM vs the 3 x bf16-split GEMM on the training-step shapes (M = 3200 tokens) and the scoring product:
median us per launch over interleaved rounds.   usage: python tools/b3_sweep.py   (on the GPU box)"""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pixelrec_amd import ops

dev = "cuda"


def timeit(fn, n):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def main():
    res = {}
    shapes = [("fwd_qkv", 3200, 1536, 512, ops.EPI_BIAS), ("fwd_o", 3200, 512, 512, ops.EPI_BIAS),
              ("fwd_f1", 3200, 1024, 512, ops.EPI_BIAS_GELU_GRAD), ("fwd_f2", 3200, 512, 1024, ops.EPI_BIAS),
              ("dx_f2", 3200, 1024, 512, ops.EPI_MUL), ("dx_f1", 3200, 512, 1024, ops.EPI_ADD),
              ("dx_o", 3200, 512, 512, ops.EPI_NONE), ("dx_qkv", 3200, 512, 1536, ops.EPI_ADD),
              ("B512_fwd_qkv", 25600, 1536, 512, ops.EPI_BIAS), ("scoring", 1024, 400001, 512, ops.EPI_NONE)]
    only = os.environ.get("SHAPES")
    for name, M, N, K, epi in shapes:
        if only and name not in only.split(","):
            continue
        x = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) * 0.05
        b = torch.randn(N, device=dev)
        big = name == "scoring"
        y = torch.empty(M, N, device=dev)
        epi = ops.EPI_BIAS
        dy = None if big else torch.randn(M, N, device=dev)
        dx = None if big else torch.empty(M, K, device=dev)
        mk = lambda th: (lambda: ops.gemm(True, True, M, N, K, x, K, W, K, y, N, epi, bias=b, use_ws=False, tile_hint=th))
        mkx = lambda th: (lambda: ops.gemm(True, False, M, K, N, dy, N, W, K, dx, K, ops.EPI_NONE, use_ws=False, tile_hint=th))
        cfgs = {"f32_64": mk(64), "f32_1281": mk(1281), "b3_64": mk(9064), "b3_1281": mk(91281), "default": mk(0)}
        if not big:
            cfgs.update({"dx_f32_64": mkx(64), "dx_b3_64": mkx(9064), "dx_b3_1281": mkx(91281)})
        samples = {k: [] for k in cfgs}
        for rnd in range(3 if big else 5):
            for k, fn in cfgs.items():
                samples[k].append(timeit(fn, 6 if big else 60))
        row = {k: round(sorted(v)[len(v) // 2], 2) for k, v in samples.items()}
        fl = 2.0 * M * N * K
        row["tflops_f32"] = round(fl / (min(row["f32_64"], row["f32_1281"]) * 1e-6) / 1e12, 1)
        row["tflops_b3_best"] = round(fl / (min(row["b3_64"], row["b3_1281"]) * 1e-6) / 1e12, 1)
        res[name] = row
        print(name, row, flush=True)
        del x, W, y, dy, dx
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "b3_sweep.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
