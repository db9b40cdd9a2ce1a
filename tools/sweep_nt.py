import os, subprocess, sys
if len(sys.argv) > 1:
    import torch
    from pixelrec_amd import ops
    N, D = 400001, 512
    t = torch.randn(N, D, device="cuda"); m = torch.zeros_like(t); v = torch.zeros_like(t)
    slot = torch.full((N,), -1, dtype=torch.int32, device="cuda")
    f = lambda: ops.adamw_table(t, m, v, slot, None, 1e-4, 0.9, 0.999, 1e-8, 0.1, 1)
    for _ in range(5): f()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(40): f()
    e.record(); torch.cuda.synchronize()
    dt = s.elapsed_time(e) / 40 * 1e-3
    print(f"PXR_ADAMW_NT={os.environ['PXR_ADAMW_NT']}: {(24.0 * N * D + 4 * N) / dt / 1e12:.3f} TB/s  {dt * 1e6:.0f} us")
else:
    for rep in range(2):
        for nt in (0, 1, 2, 3):
            subprocess.run([sys.executable, __file__, "x"], env=dict(os.environ, PXR_ADAMW_NT=str(nt), PYTHONPATH="."))
