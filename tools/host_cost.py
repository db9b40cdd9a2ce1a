"""Host-side cost per step: tiny shapes so the GPU is never the bottleneck."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from pixelrec_amd import synth
from pixelrec_amd.model import SASRec
from pixelrec_amd.optim import PxrAdamW
cfg = {"n_layers": 2, "n_heads": 4, "embedding_size": 64, "inner_size": 2, "hidden_dropout_prob": 0.1, "attn_dropout_prob": 0.1,
       "hidden_act": "gelu", "layer_norm_eps": 1e-12, "initializer_range": 0.02, "MAX_ITEM_LIST_LENGTH": 10, "seed": 1}
class DL: item_num = 2000
m = SASRec(cfg, DL()).cuda().train(); opt = PxrAdamW(m)
rng = np.random.default_rng(0)
it, mk = synth.train_batch(2000, 8, 10, rng, synth.ZipfItems(2000))
it, mk = torch.from_numpy(it).cuda(), torch.from_numpy(mk).cuda()
def timeit(fn, n=300):
    for _ in range(20): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
def fwd():
    with torch.no_grad():
        m._ensure_packed(); m.training = True; m._forward_train(it, mk)
def fwdbwd(): m((it, mk)).backward()
def full(): m((it, mk)).backward(); opt.step()
print("fwd only      %.3f ms" % timeit(fwd))
for ov in (True, False):
    m.overlap_weight_grads = ov
    print("overlap=%s fwd+bwd %.3f ms   full step %.3f ms" % (ov, timeit(fwdbwd), timeit(full)))
