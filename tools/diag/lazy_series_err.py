"""Where the series replay of the lazy table AdamW (csrc/adamw.hip) stands against the exact replay and the carried-product loop:
(1) the 700-step optimizer-only harness of tests/test_gpu_lazy_adamw.py in the three modes, (2) one catch-up at several optimizer
steps, error relative to the size of the summed update.  python tools/diag/lazy_series_err.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import test_gpu_lazy_adamw as H   # noqa: E402
import test_gpu_lazy_series as S  # noqa: E402


def mode(name):
    os.environ.pop("PXR_LAZY_REPLAY", None)
    os.environ.pop("PXR_LAZY_SERIES", None)
    if name == "exact":
        os.environ["PXR_LAZY_REPLAY"] = "exact"
    elif name == "loop":
        os.environ["PXR_LAZY_SERIES"] = "0"


class MP:
    def setenv(self, k, v): os.environ[k] = v
    def delenv(self, k, raising=False): os.environ.pop(k, None)


for T in (700, 1500):
    for name in ("exact", "loop", "series"):
        mode(name)
        (pd, md, vd), (pl, ml, vl), p64 = H._optimizer_only_run(T=T, truth=True)
        print(f"harness T={T} {name:6s} max|p_dense - p_lazy| = {(pd - pl).abs().max().item():.3e}   vs float64: lazy max {(pl.double() - p64).abs().max().item():.3e} mean {(pl.double() - p64).abs().mean().item():.3e} | dense max {(pd.double() - p64).abs().max().item():.3e} mean {(pd.double() - p64).abs().mean().item():.3e}")
for T in (300, 700, 1500, 5000):
    hyper, cumlog = S._table(T, lambda t: 1e-3)
    gaps = [8, 16, 31, 64, 100, 128]
    state = S._rows(1 + 6 * 32, 512, gaps, T, seed=T)
    p0 = state[0][1:]
    ex = S._catch_up(MP(), "exact", state, hyper, cumlog, T)
    for name in ("loop", "series"):
        out = S._catch_up(MP(), name, state, hyper, cumlog, T)
        upd = (ex[0] - p0).abs()
        err = (out[0] - ex[0]).abs()
        rel = (err / upd.clamp_min(1e-4)).max().item()
        print(f"catch-up T={T} {name:6s} max abs err {err.max().item():.3e}  max err / max(update, 1e-4) {rel:.3e}  max update {upd.max().item():.3e}")


print("against the float64 recurrence (gaps <= 128: inside every mode's window):")
for T in (300, 700, 1500, 5000):
    hyper, cumlog = S._table(T, lambda t: 1e-3)
    gaps = [8, 16, 31, 64, 100, 128]
    state = S._rows(1 + 6 * 32, 512, gaps, T, seed=T)
    tr = S._truth64(state, hyper, T)
    for name in ("exact", "loop", "series"):
        out = S._catch_up(MP(), name, state, hyper, cumlog, T)
        err = (out[0].double() - tr).abs()
        print(f"  T={T} {name:6s} max |p - p64| {err.max().item():.3e}   mean {err.mean().item():.3e}")
