import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["PXR_SEQ_H2_STALE"] = "0"
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import test_gpu_schedule_ab as T
class MP:
    def setenv(self, k, v): os.environ[k] = v
for steps in (1, 2, 3, 4, 5, 6, 7, 9):
    T.STEPS = steps
    e = T._run(MP(), "1", "1", False); g = T._run(MP(), "1", "1", True)
    bad = {k: float((e[1][k] - g[1][k]).abs().max()) for k in e[1] if not torch.equal(e[1][k], g[1][k])}
    print("STEPS", steps, "losses equal", e[0] == g[0], "differing keys", len(bad), list(bad.items())[:3], flush=True)
