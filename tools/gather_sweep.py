"""Sweeps PXR_GATHER_VARIANT (one process per variant: the library reads it once)."""
import os
import subprocess
import sys

if len(sys.argv) > 1:
    import torch

    from pixelrec_amd import ops

    N, D, L = 400001, 512, 50
    n_rows = 2048 * 2 * (L + 1)
    table = torch.randn(N, D, device="cuda")
    for name, idx in (("uniform", torch.randint(1, N, (n_rows,), device="cuda")),):
        for _ in range(5):
            ops.embed_gather(table, idx)
        torch.cuda.synchronize()
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        for _ in range(50):
            ops.embed_gather(table, idx)
        e.record()
        torch.cuda.synchronize()
        t = s.elapsed_time(e) / 50 * 1e-3
        print(f"variant {os.environ.get('PXR_GATHER_VARIANT')}: {name} {2.0 * n_rows * D * 4 / t / 1e12:.3f} TB/s ({t * 1e6:.1f} us)")
else:
    for v in range(12):
        subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, PXR_GATHER_VARIANT=str(v), PYTHONPATH="."))
