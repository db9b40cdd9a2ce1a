import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from tests.test_gpu_lazy_adamw import _setup, _run
make, rng, zipf, synth = _setup(n_items=1500)
nb = int(os.environ.get("NB", 60))
batches = [tuple(torch.from_numpy(x).cuda() for x in synth.train_batch(1500, 4, 10, rng, zipf)) for _ in range(nb)]
dense, od = _run(make, batches, "dense")
lazy, ol = _run(make, batches, "lazy")
for k in dense:
    d = (dense[k] - lazy[k]).abs()
    if d.max() > 0:
        print(k, "maxdiff", d.max().item(), "n_diff_elems", int((d > 0).sum()), "of", d.numel())
t = (dense["item_embedding.weight"] - lazy["item_embedding.weight"]).abs().max(1).values
rows = torch.nonzero(t > 0).squeeze(1)
print("differing rows:", rows[:20].tolist(), len(rows))
touched = set()
for it, mk in batches:
    touched |= set(it.flatten().tolist())
print("of which touched:", sum(int(r) in touched for r in rows.tolist()), "untouched:", sum(int(r) not in touched for r in rows.tolist()))
print("m diff", (od._tm - ol._tm).abs().max().item(), "v diff", (od._tv - ol._tv).abs().max().item())
