"""TEST INFRASTRUCTURE ONLY -- makes the *reference* (read-only, /root/reference) importable in the dev
container so that golden vectors can be generated from it (oracle/make_golden.py).  It never travels to the
GPU box and nothing in pixelrec_amd/ imports it.

The reference imports six third-party packages that are absent here and are NOT on the SASRec arithmetic path
(SURVEY.md §8c): torch_geometric (layers.py:9-10, dataload.py:14), tensorboardX (utils/utils.py:8), colorlog
and colorama (utils/logger.py:3,7), lmdb and torchvision (data/dataset/trainset.py:6-8, batchset.py:7-9).
They are replaced by inert `sys.modules` stubs; no reference source is copied or modified.
"""
import os
import sys
import types

REFERENCE_ROOT = os.environ.get("PXR_REFERENCE_ROOT", "/root/reference")
REFERENCE_CODE = os.path.join(REFERENCE_ROOT, "code")


def reference_available() -> bool:
    return os.path.isdir(os.path.join(REFERENCE_CODE, "REC"))


def _mod(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


def install_stubs():
    import torch.nn as nn

    sys.dont_write_bytecode = True  # the reference tree is read-only

    class _MessagePassing(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    def _unavailable(*a, **k):
        raise RuntimeError("stubbed third-party function (not on the SASRec path)")

    if "torch_geometric" not in sys.modules:
        tg = _mod("torch_geometric")
        tg.nn = _mod("torch_geometric.nn", MessagePassing=_MessagePassing)
        tg.utils = _mod("torch_geometric.utils", add_self_loops=_unavailable, degree=_unavailable)

    class _SummaryWriter:
        def __init__(self, *a, **k):
            pass

        def add_scalar(self, *a, **k):
            pass

        def add_hparams(self, *a, **k):
            pass

    if "tensorboardX" not in sys.modules:
        _mod("tensorboardX", SummaryWriter=_SummaryWriter)

    import logging

    class _ColoredFormatter(logging.Formatter):
        def __init__(self, fmt=None, datefmt=None, log_colors=None, **k):
            fmt = (fmt or "%(message)s").replace("%(log_color)s", "")
            super().__init__(fmt, datefmt)

    if "colorlog" not in sys.modules:
        _mod("colorlog", ColoredFormatter=_ColoredFormatter)
    if "colorama" not in sys.modules:
        _mod("colorama", init=lambda *a, **k: None)
    if "lmdb" not in sys.modules:
        _mod("lmdb", open=_unavailable)
    if "torchvision" not in sys.modules:
        tv = _mod("torchvision")
        tv.transforms = _mod("torchvision.transforms", Compose=_unavailable, Resize=_unavailable,
                             ToTensor=_unavailable, Normalize=_unavailable)
        tv.models = _mod("torchvision.models")
    if "clip" not in sys.modules:          # OpenAI CLIP package: imported by REC/model/load.py:2, unused by the ViT branch
        _mod("clip", load=_unavailable)
    for name in ("wandb", "hyperopt"):
        if name not in sys.modules:
            try:
                __import__(name)
            except Exception:
                _mod(name)


def import_reference():
    """Returns the reference `REC` package (imported from /root/reference/code, unmodified)."""
    if not reference_available():
        raise RuntimeError(f"reference not found under {REFERENCE_ROOT}")
    install_stubs()
    if REFERENCE_CODE not in sys.path:
        sys.path.insert(0, REFERENCE_CODE)
    import REC  # noqa: F401

    return sys.modules["REC"]
