"""Stream-K vs tile-per-workgroup for the GEMM shapes of the training step (M = B*L = 3200 tokens): avg us per launch.
usage: python tools/sk_sweep.py   (on the GPU box)"""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pixelrec_amd import ops

dev = "cuda"


def timeit(fn, n=200):
    for _ in range(20):
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
    M = 3200
    res = {}
    shapes = [("fwd_qkv KC,KC bias", True, 1536, 512, ops.EPI_BIAS), ("fwd_o KC,KC bias", True, 512, 512, ops.EPI_BIAS),
              ("fwd_f1 KC,KC gelu", True, 1024, 512, ops.EPI_BIAS_GELU_GRAD), ("fwd_f2 KC,KC bias", True, 512, 1024, ops.EPI_BIAS),
              ("dx_f2 KC,XC mul", False, 1024, 512, ops.EPI_MUL), ("dx_f1 KC,XC add", False, 512, 1024, ops.EPI_ADD),
              ("dx_o KC,XC", False, 512, 512, ops.EPI_NONE), ("dx_qkv KC,XC add", False, 512, 1536, ops.EPI_ADD)]
    for name, bkc, N, K, epi in shapes:
        x = torch.randn(M, K, device=dev)
        W = torch.randn(N, K, device=dev) if bkc else torch.randn(K, N, device=dev)
        b = torch.randn(N, device=dev)
        aux = torch.randn(M, N, device=dev)
        y = torch.empty(M, N, device=dev)
        cfgs = (("tile64", 64, 0), ("sk512", 6464, 512), ("sk768", 6464, 768), ("sk1024", 6464, 1024), ("sk400", 6464, 400),
                ("default", 0, 0))
        samples = {c[0]: [] for c in cfgs}
        for rnd in range(5):                 # interleaved rounds: the clock state drifts within a measurement series
            for label, th, sh in cfgs:
                fn = lambda: ops.gemm(True, bkc, M, N, K, x, K, W, K if bkc else N, y, N, epi, bias=b, aux=aux, ldaux=N,
                                      use_ws=False, tile_hint=th, split_hint=sh)
                samples[label].append(timeit(fn, 60))
        row = {k: round(sorted(v)[len(v) // 2], 2) for k, v in samples.items()}
        fl = 2.0 * M * N * K
        row["tflops_best"] = round(fl / (min(row.values()) * 1e-6) / 1e12, 1)
        res[name] = row
        print(name, row, flush=True)
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "sk_sweep.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
