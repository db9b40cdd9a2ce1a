"""Helpers to read tests/golden/sasrec_*.npz (written by oracle/make_golden.py from the reference)."""
import os

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
META_KEYS = ("n_items", "D", "L", "H", "inner", "n_layers", "B", "seed")
CASES = ("tiny", "cfg1", "ns")


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, f"sasrec_{name}.npz"), allow_pickle=False)
    meta = dict(zip(META_KEYS, [int(x) for x in z["meta"]]))
    return meta, z


def oracle_cfg(meta, train_dropout=False):
    return {"n_layers": meta["n_layers"], "n_heads": meta["H"], "layer_norm_eps": 1e-12,
            "hidden_dropout_prob": 0.1 if train_dropout else 0.0,
            "attn_dropout_prob": 0.1 if train_dropout else 0.0}


def compare(z, name, t, atol, rtol=0.0, what=""):
    """Compare tensor/array t with the golden entry `name` (full or strided sub-sample + sum + l2)."""
    a = t.detach().cpu().numpy() if isinstance(t, torch.Tensor) else np.asarray(t)
    if name in z.files:
        ref = z[name]
        err = np.abs(a.astype(np.float64) - ref.astype(np.float64))
        tol = atol + rtol * np.abs(ref)
        assert a.shape == ref.shape, f"{what}{name}: shape {a.shape} vs {ref.shape}"
        assert (err <= tol).all(), f"{what}{name}: max err {err.max():.3e} > tol (atol {atol:g}, rtol {rtol:g})"
        return float(err.max())
    stride = int(z[name + "@stride"])
    ref = z[name + "@sub"]
    sub = a.reshape(-1)[::stride]
    err = np.abs(sub.astype(np.float64) - ref.astype(np.float64))
    tol = atol + rtol * np.abs(ref)
    assert (err <= tol).all(), f"{what}{name}@sub: max err {err.max():.3e}"
    l2 = float(z[name + "@l2"])
    got = float(np.sqrt((a.astype(np.float64) ** 2).sum()))
    assert abs(got - l2) <= 1e-4 * max(l2, 1e-12) + atol, f"{what}{name}@l2: {got} vs {l2}"
    return float(err.max())
