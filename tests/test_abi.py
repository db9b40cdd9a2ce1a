"""The C-ABI library loads and exports exactly what include/pxr.h declares (no compute calls: CPU only)."""
import ctypes
import os
import re

from pixelrec_amd import lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _header_symbols():
    text = open(os.path.join(ROOT, "include", "pxr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pxr_[a-z0-9_]+)\s*\(", text)))


def test_library_loads_and_reports_target():
    L = lib.load()
    assert L.pxr_version() >= 100
    assert L.pxr_target_arch() == b"gfx950"


def test_header_and_binding_agree():
    assert _header_symbols() == lib.exported_symbols()


def test_every_declared_symbol_is_exported():
    raw = ctypes.CDLL(lib.LIB_PATH)
    for name in _header_symbols():
        assert hasattr(raw, name), f"libpxr.so does not export {name}"


def test_bad_arguments_fail_loudly_without_a_gpu():
    L = lib.load()
    # null pointers are rejected before any launch, with a message
    rc = L.pxr_embed_gather_f32(None, 10, 8, None, 4, None, None)
    assert rc == -1
    assert b"null pointer" in L.pxr_last_error()
    assert L.pxr_score_topk_ws_bytes(8, 100, 64) == -1     # K > 32


def test_no_cpu_fallback():
    import pytest
    import torch

    from pixelrec_amd import ops

    with pytest.raises(lib.PxrError):
        ops.embed_gather(torch.zeros(4, 8), torch.zeros(2, dtype=torch.int64))
