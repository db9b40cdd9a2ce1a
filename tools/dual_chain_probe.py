"""Would two half-batch chains on two HIP streams beat one full-batch chain?  (B = 64 per GPU is ~50 dependent kernels of
~20 us, each with ~5 us of ramp / tail where the matrix pipes idle.)  Probe without refactoring the engine: two independent
SASRec replicas at B/2 captured as two parallel branches of ONE hipGraph vs one replica at B.  Prints us per step.
usage: python tools/dual_chain_probe.py   (on the GPU box)"""
import json
import os
import sys
import time

sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np
import torch

import bench
from pixelrec_amd import synth
from pixelrec_amd.model import SASRec
from pixelrec_amd.optim import PxrAdamW

dev = torch.device("cuda", 0)
NS = bench.NS
N, L = NS["n_items"], NS["L"]


class DL:
    item_num = N


def make(B, seed):
    torch.manual_seed(seed)
    with torch.device(dev):
        m = SASRec(bench.model_config(0.1), DL())
    m.train()
    m.defer_weight_grad_join = True
    o = PxrAdamW(m, lr=1e-4, weight_decay=0.1, table_update="lazy")
    rng = np.random.default_rng(seed)
    zipf = synth.ZipfItems(N, seed=2020)
    batches = [tuple(torch.from_numpy(a).to(dev) for a in synth.train_batch(N, B, L, rng, zipf)) for _ in range(64)]
    return m, o, batches


def eager(m, o, b, one):
    o.zero_grad()
    loss = m(b)
    loss.backward(one)
    o.step()
    return loss


def capture(chains, n_streams):
    """chains: list of (model, opt, static items, static mask).  One graph; chain i runs on stream i % n_streams."""
    one = torch.ones((), dtype=torch.float32, device=dev)
    cur = torch.cuda.current_stream()
    side = [torch.cuda.Stream() for _ in range(n_streams)]
    for s in side:
        s.wait_stream(cur)
    for i, (m, o, it, mk) in enumerate(chains):
        with torch.cuda.stream(side[i % n_streams]):
            for _ in range(3):
                eager(m, o, (it, mk), one)
    for s in side:
        cur.wait_stream(s)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    cs = torch.cuda.Stream()
    cs.wait_stream(cur)
    with torch.cuda.stream(cs):
        with torch.cuda.graph(g, stream=cs):
            for s in side:
                s.wait_stream(cs)
            for i, (m, o, it, mk) in enumerate(chains):
                with torch.cuda.stream(side[i % n_streams]):
                    eager(m, o, (it, mk), one)
            for s in side:
                cs.wait_stream(s)
    cur.wait_stream(cs)
    return g


def time_graph(g, chains, pools, n=60):
    for _ in range(10):
        g.replay()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(n):
        for (m, o, it, mk), pool in zip(chains, pools):
            b = pool[k % len(pool)]
            it.copy_(b[0], non_blocking=True); mk.copy_(b[1], non_blocking=True)
        g.replay()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


def main():
    B = 64
    res = {}
    m, o, pool = make(B, 1)
    chain = [(m, o, pool[0][0].clone(), pool[0][1].clone())]
    g = capture(chain, 1)
    res["one_chain_B64_us"] = time_graph(g, chain, [pool])
    del g, chain, m, o, pool
    torch.cuda.empty_cache()
    a = make(B // 2, 2)
    b = make(B // 2, 3)
    chains = [(a[0], a[1], a[2][0][0].clone(), a[2][0][1].clone()), (b[0], b[1], b[2][0][0].clone(), b[2][0][1].clone())]
    g2 = capture(chains, 2)
    res["two_chains_B32_two_streams_us"] = time_graph(g2, chains, [a[2], b[2]])
    g1 = capture(chains, 1)
    res["two_chains_B32_one_stream_us"] = time_graph(g1, chains, [a[2], b[2]])
    del g1, g2, chains, a, b
    torch.cuda.empty_cache()
    qs = [make(B // 4, 10 + i) for i in range(4)]
    chains = [(q[0], q[1], q[2][0][0].clone(), q[2][0][1].clone()) for q in qs]
    g4 = capture(chains, 4)
    res["four_chains_B16_four_streams_us"] = time_graph(g4, chains, [q[2] for q in qs])
    res["speedup_two_streams"] = res["one_chain_B64_us"] / res["two_chains_B32_two_streams_us"]
    res["speedup_four_streams"] = res["one_chain_B64_us"] / res["four_chains_B16_four_streams_us"]
    print(json.dumps(res, indent=1))
    out = os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "gpurun_out", "dual_chain_probe.json")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    json.dump(res, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
