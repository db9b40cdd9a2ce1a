import torch, numpy as np, os
from pixelrec_amd import ops, synth
N, B, L = 400001, 64, 50
rng = np.random.default_rng(0)
it, _ = synth.train_batch(N, B, L, rng, synth.ZipfItems(N, seed=1))
items = torch.from_numpy(it).cuda()
from pixelrec_amd import lib as _l
Lb = _l.load()
n = 3 * B * L
ws_bytes = int(Lb.pxr_embed_grad_ws_bytes(n))
ws = torch.empty(ws_bytes, dtype=torch.uint8, device="cuda")
sp = ops.SparseRows(B * (2 * L + 1), 512, "cuda")
def f():
    _l.check(Lb.pxr_sasrec_occ_sort(_l.ptr(items), B, L, N, _l.ptr(sp.idx), _l.ptr(sp.n), _l.ptr(ws), ws_bytes, _l.stream_ptr()), "x")
for _ in range(5): f()
torch.cuda.synchronize()
s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
s.record()
for _ in range(200): f()
e.record(); torch.cuda.synchronize()
print("fused sort:", os.environ.get("PXR_FUSED_SORT", "1"), f"{s.elapsed_time(e) / 200 * 1e3:.1f} us/call")
