"""Where the time of a forward GEMM of the B = 64 step goes: the h2 GEMMs (3 200 tokens) replayed from a hipGraph with parts of the
main loop switched off (PXR_P3_DBG, timing only -- results are wrong): 2 = no DMA, 4 = no fragment reads / MFMAs, 8 = no barriers.
python tools/diag/gemm_dbg_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from pixelrec_amd import ops  # noqa: E402

dev = "cuda"


def graph_time(fn, n=32):
    fn(); fn()
    torch.cuda.synchronize()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for i in range(n):
                fn()
    torch.cuda.current_stream().wait_stream(st)
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    return sorted(ts)[3]


res = {}
for name, M, N, K in (("qkv", 3200, 1536, 512), ("out", 3200, 512, 512), ("fc1", 3200, 1024, 512), ("fc2", 3200, 512, 1024)):
    x = torch.randn(M, K, device=dev)
    W = torch.randn(N, K, device=dev) * 0.05
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    xp, = ops.split_h2_auto([x])
    Wp, = ops.split_h2_auto([W])
    row = {}
    hint = int(os.environ.get("PROBE_TILE", "0"))
    for dbg in (0, 2, 4, 8, 6, 14, 16, 32, 6 + 16, 6 + 32):
        os.environ["PXR_P3_DBG"] = str(dbg)
        row[f"dbg{dbg}"] = round(graph_time(lambda: ops.gemm_planes(xp, Wp, y, ops.EPI_BIAS, bias=b, tile_hint=hint)), 2)
    os.environ.pop("PXR_P3_DBG")
    res[name] = row
    print(name, json.dumps(row), flush=True)
e = torch.empty(1, device=dev)
print("empty-ish launch (fill of one float):", round(graph_time(lambda: e.fill_(1.0)), 2), "us")
json.dump(res, open("gpurun_out/gemm_dbg_probe.json", "w"), indent=1)
