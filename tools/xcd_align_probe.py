"""Does a GEMM run faster when the XCD that reads an A-operand row range is the XCD that WROTE it (operand still in that
XCD's L2)?  [split(A) ; gemm] pairs replayed from a hipGraph with the split kernel's rows XCD-aligned or not.
usage: python tools/xcd_align_probe.py (GPU box)"""
import json
import os
import sys

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from pixelrec_amd import ops

dev = "cuda"


def graph_time(fn, n=20):
    fn(); torch.cuda.synchronize()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for _ in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(st)
    ts = []
    for _ in range(5):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    return sorted(ts)[2]


res = {}
for name, M, N, K in (("o", 3200, 512, 512), ("f2", 3200, 512, 1024), ("qkv", 3200, 1536, 512)):
    x = torch.randn(M, K, device=dev); W = torch.randn(N, K, device=dev) * 0.05; b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    xp, Wp = ops.split_planes(x), ops.split_planes(W)
    trash = torch.empty(64 << 20, device=dev)        # 256 MB written between the pairs: evicts L2 and most of the MALL
    row = {}
    for mode in ("1", "0"):
        os.environ["PXR_SPLIT_XCD"] = mode
        pair = lambda: (ops.split_planes(x, xp), ops.gemm_planes(xp, Wp, y, ops.EPI_BIAS, bias=b))
        only = lambda: ops.split_planes(x, xp)
        row[f"split_xcd{mode}_us"] = round(graph_time(only), 2)
        row[f"pair_xcd{mode}_us"] = round(graph_time(pair), 2)
        row[f"gemm_after_split_xcd{mode}_us"] = round(row[f"pair_xcd{mode}_us"] - row[f"split_xcd{mode}_us"], 2)
    row["gemm_alone_hot_us"] = round(graph_time(lambda: ops.gemm_planes(xp, Wp, y, ops.EPI_BIAS, bias=b)), 2)
    res[name] = row
    print(name, json.dumps(row), flush=True)
json.dump(res, open(os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "xcd_align_probe.json"), "w"), indent=1)
