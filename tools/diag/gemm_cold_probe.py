"""Why a forward / input-gradient GEMM of the B = 64 step takes 20-30 % longer inside the step than back to back in a loop: the same
h2 GEMM (3 200 tokens) replayed from a hipGraph (a) on one weight matrix (its planes stay in every XCD's L2), (b) rotating over 16 /
128 distinct weight matrices (L2-cold, MALL-resident / past the MALL: the step touches each weight once per pass), (c) rotating the
activations too, (d) with the activations written right before by the plane split (their producer in the step).
python tools/diag/gemm_cold_probe.py"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402
from pixelrec_amd import ops  # noqa: E402

dev = "cuda"


def graph_time(fns, n=32):
    for f in fns[:2]:
        f()
    torch.cuda.synchronize()
    st = torch.cuda.Stream(); st.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(st):
        with torch.cuda.graph(g, stream=st):
            for i in range(n):
                fns[i % len(fns)]()
    torch.cuda.current_stream().wait_stream(st)
    ts = []
    for _ in range(7):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); g.replay(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b) / n * 1e3)
    return sorted(ts)[3]


res = {}
for name, M, N, K in (("qkv", 3200, 1536, 512), ("out", 3200, 512, 512), ("fc1", 3200, 1024, 512), ("fc2", 3200, 512, 1024)):
    xs = [torch.randn(M, K, device=dev) for _ in range(16)]
    Ws = [torch.randn(N, K, device=dev) * 0.05 for _ in range(128)]
    b = torch.randn(N, device=dev)
    y = torch.empty(M, N, device=dev)
    xps = ops.split_h2_auto(xs)
    Wps = []
    for lo in range(0, 128, 16):
        Wps += ops.split_h2_auto(Ws[lo:lo + 16])
    g = lambda xp, Wp: (lambda: ops.gemm_planes(xp, Wp, y, ops.EPI_BIAS, bias=b))
    row = {"hot": graph_time([g(xps[0], Wps[0])]),
           "w_rot16": graph_time([g(xps[0], Wps[i]) for i in range(16)]),
           "w_rot128": graph_time([g(xps[0], Wps[i]) for i in range(128)], n=128),
           "x_rot16": graph_time([g(xps[i], Wps[0]) for i in range(16)]),
           "both_rot16": graph_time([g(xps[i], Wps[i]) for i in range(16)])}
    # activations written by their producer right before (a split launch per GEMM, timed alone and subtracted)
    outs = ops.split_h2_auto([xs[0]], with_buffers=True)
    sp = lambda: ops.split_h2_auto([xs[0]], outs=outs)
    t_sp = graph_time([sp])
    pair = [(lambda i=i: (sp(), ops.gemm_planes(outs[0][0], Wps[i], y, ops.EPI_BIAS, bias=b))) for i in range(16)]
    row["after_split_w_rot16"] = graph_time(pair) - t_sp
    res[name] = {k: round(v, 2) for k, v in row.items()}
    print(name, json.dumps(res[name]), flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(res, open("gpurun_out/gemm_cold_probe.json", "w"), indent=1)
