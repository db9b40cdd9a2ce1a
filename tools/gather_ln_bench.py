"""The embedding gather AS THE STEP RUNS IT -- ln_fwd_kernel<GATHER>: table rows + position rows -> LayerNorm -> dropout -> y, xhat
and the planes of y -- timed alone at the step's batch sizes, uniform and Zipf ids, under the store policies of PXR_LN_NT
(bit 0 y, 1 planes, 2 xhat, 3 row loads; each setting in a fresh process because the library reads the knob once).
usage: python tools/gather_ln_bench.py [nt ...]   -> one JSON line per (nt, B, ids, planes)"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def child(nt):
    import numpy as np
    import torch

    from pixelrec_amd import ops, synth

    dev = torch.device("cuda:0")
    N, D, L = 400_001, 512, 50
    table = torch.randn(N, D, device=dev) * 0.02
    pos = torch.randn(L, D, device=dev) * 0.02
    g, b = torch.ones(D, device=dev), torch.zeros(D, device=dev)
    zipf = synth.ZipfItems(N, seed=2020)
    rng = np.random.default_rng(0)
    for B in (64, 512, 2048):
        for ids in ("uniform", "zipf"):
            if ids == "uniform":
                idx = torch.randint(1, N, (B, L), device=dev)
            else:
                idx = torch.from_numpy(synth.train_batch(N, B, L, rng, zipf)[0][:, 0, :L].copy()).to(dev)
            for planes, p_drop, save in (("h2", 0.1, True), (True, 0.1, True), ("h2", 0.0, True), ("h2", 0.1, False), (False, 0.1, True),
                                         (False, 0.0, False)):
                if (p_drop, save) != (0.1, True) and (nt != -1 or ids != "uniform"):
                    continue          # the decomposition runs (dropout off / fewer outputs) once, under the default policy
                fn = lambda: ops.input_ln_fwd(table, idx, L, B, L, pos, g, b, 1e-12, p_drop, 12345, 0, save=save, planes=planes)
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                iters = 20
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(iters):
                    fn()
                e.record()
                torch.cuda.synchronize()
                t = s.elapsed_time(e) / iters * 1e-3
                by = B * L * D * (4.0 * (3 if save else 2) + (4.0 if planes == "h2" else 6.0 if planes else 0.0))
                print(json.dumps({"nt": nt, "B": B, "ids": ids, "planes": "h2" if planes == "h2" else "bf16x3" if planes else "none",
                                  "p_drop": p_drop, "xhat": save, "us": t * 1e6,
                                  "GBps": by / t / 1e9, "frac_of_8TBps": by / t / 8e12,
                                  "bytes": "rows read + y + xhat (fp32) + planes of y"}), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(int(sys.argv[2]))
    else:
        for nt in (sys.argv[1:] or ["-1", "0", "4", "7", "15", "3", "5"]):
            env = dict(os.environ)
            if int(nt) >= 0:
                env["PXR_LN_NT"] = nt
            else:
                env.pop("PXR_LN_NT", None)
            subprocess.call([sys.executable, os.path.abspath(__file__), "--child", nt], env=env)
