"""Diagnostic: run the bench's B=64 step eagerly with the stale-scale gradient planes and print, per step, each site's exponent in use
against this step's true maximum (from the partial maxima) -- how much of the headroom a step-to-step change really takes."""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from pixelrec_amd import ops, synth
from pixelrec_amd.model import SASRec
from pixelrec_amd.optim import PxrAdamW
import bench
torch.manual_seed(2020)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
N, L, D = 400_001, 50, 512
class DL: item_num = N
with torch.device("cuda"):
    m = SASRec(bench.model_config(0.1), DL())
m.train(); m.h2_stale_scales = True; m.trust_optimizer_planes = True
opt = PxrAdamW(m, lr=1e-4, weight_decay=0.1)
rng = np.random.default_rng(int(os.environ.get("WATCH_SEED", "2020"))); zipf = synth.ZipfItems(N, seed=2020)
orig = ops.H2Sites.update
log = []
def upd(self, parts, n_parts, rows, bound_b, f):
    mx = [float(p[:n].max()) for p, n in zip(parts, n_parts)]
    used = self.exps.tolist()
    log.append((mx, used))
    return orig(self, parts, n_parts, rows, bound_b, f)
ops.H2Sites.update = upd
graph = len(sys.argv) > 3 and sys.argv[3] == "graph"
if graph:
    ops.H2Sites.update = orig
    from pixelrec_amd.graph import GraphedTrainStep
    from pixelrec_amd.parallel import DataParallel
    b0 = synth.train_batch(N, B, L, rng, zipf)
    g = GraphedTrainStep(DataParallel(m), opt, torch.from_numpy(b0[0]).cuda(), torch.from_numpy(b0[1]).cuda())
    prev = None
    for s in range(steps):
        it, mk = synth.train_batch(N, B, L, rng, zipf)
        loss = g(torch.from_numpy(it).cuda(), torch.from_numpy(mk).cuda())
        torch.cuda.synchronize()
        st = int(ops.device_status("cuda").item())
        ex = m._h2_sites.exps.tolist()
        if st or s < 5 or s % 50 == 0 or ex != prev:
            print(f"replay {s} loss {float(loss):.4f} status {st} exps {ex} stats {[f'{x:.2e}' for x in m._h2_sites.stats[:,0].tolist()]} bexp {m._h2_sites.bexp.tolist()}", flush=True)
        prev = ex
        if st:
            break
    sys.exit(0)
for s in range(steps):
    it, mk = synth.train_batch(N, B, L, rng, zipf)
    loss = m((torch.from_numpy(it).cuda(), torch.from_numpy(mk).cuda()))
    loss.backward(); opt.step()
    mx, used = log[-1]
    top = [math.log2(x) + e if x > 0 else float("nan") for x, e in zip(mx, used)]
    st = int(ops.device_status("cuda").item())
    print(f"step {s} loss {float(loss):.4f} status {st} log2(max*2^exp) per site:", " ".join(f"{t:5.1f}" for t in top), " max:", " ".join(f"{x:.2e}" for x in mx), flush=True)
