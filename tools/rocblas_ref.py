import torch, os
dev = torch.device("cuda:0")
def timeit(fn, iters=30, warm=5):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) / iters * 1e-3
M = 3200
for (m, n, k, kind) in [(M,1536,512,"nt"), (M,512,512,"nt"), (M,1024,512,"nt"), (M,512,1024,"nt"), (M,512,1536,"nn"), (M,512,512,"nn"), (1536,512,M,"tn"), (512,512,M,"tn"), (102400,512,512,"nt")]:
    if kind == "nt":
        a = torch.randn(m, k, device=dev); b = torch.randn(n, k, device=dev); f = lambda: torch.mm(a, b.t())
    elif kind == "nn":
        a = torch.randn(m, k, device=dev); b = torch.randn(k, n, device=dev); f = lambda: torch.mm(a, b)
    else:
        a = torch.randn(k, m, device=dev); b = torch.randn(k, n, device=dev); f = lambda: torch.mm(a.t(), b)
    t = timeit(f)
    print(f"rocBLAS {kind} M={m} N={n} K={k}: {t*1e6:7.1f} us  {2*m*n*k/t/1e12:6.1f} TF")
