import cProfile, pstats, sys, os, io
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pixelrec_amd import synth
from pixelrec_amd.model import SASRec
from pixelrec_amd.optim import PxrAdamW
import bench
class DL: item_num = 400001
m = SASRec(bench.model_config(0.1), DL()).cuda().train()
opt = PxrAdamW(m)
rng = np.random.default_rng(0); z = synth.ZipfItems(400001)
it, mk = synth.train_batch(400001, 64, 50, rng, z)
it, mk = torch.from_numpy(it).cuda(), torch.from_numpy(mk).cuda()
def step():
    loss = m((it, mk)); loss.backward(); opt.step()
for _ in range(10): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(100): step()
pr.disable(); torch.cuda.synchronize()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28); print(s.getvalue()[:6000])
