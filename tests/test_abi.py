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


def _header_prototypes():
    """name -> list of C parameter type strings, parsed from include/pxr.h."""
    text = open(os.path.join(ROOT, "include", "pxr.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(pxr_[a-z0-9_]+)\s*\(([^;{]*?)\)\s*;", text, flags=re.S):
        args = [a.strip() for a in m.group(2).replace("\n", " ").split(",")]
        protos[m.group(1)] = [] if args in ([""], ["void"]) else args
    return protos


def test_ctypes_signatures_match_the_header():
    """Every binding in lib._SIGNATURES has the header's arity and, per argument, the right ctypes class (pointer vs
    32/64-bit integer vs float/double): a drifted prototype would otherwise corrupt the call silently."""
    protos = _header_prototypes()
    for name, (_, argtypes) in lib._SIGNATURES.items():
        cargs = protos[name]
        assert len(cargs) == len(argtypes), (name, len(cargs), len(argtypes))
        for c, t in zip(cargs, argtypes):
            if "*" in c:
                want = ("c_void_p", "c_char_p", "LP_", "Array")
                assert any(w in t.__name__ for w in want) or t is ctypes.c_void_p, (name, c, t)
            elif re.search(r"\b(int64_t|uint64_t|size_t)\b", c):
                assert t in (ctypes.c_int64, ctypes.c_uint64, ctypes.c_size_t), (name, c, t)
            elif re.search(r"\bdouble\b", c):
                assert t is ctypes.c_double, (name, c, t)
            elif re.search(r"\bfloat\b", c):
                assert t is ctypes.c_float, (name, c, t)
            else:
                assert t in (ctypes.c_int, ctypes.c_int32, ctypes.c_uint32), (name, c, t)


def test_packed_block_layout_host_side():
    """The packed exchange block of include/pxr.h: sizes are host arithmetic (no GPU), and ops.SparseRows(packed=True)
    lays its three views over one buffer exactly there."""
    import torch

    from pixelrec_amd import ops

    L = lib.load()
    for cap, D in [(1, 4), (7, 8), (6464, 512), (100, 4096)]:
        off = L.pxr_packed_rows_offset(cap)
        assert off % 16 == 0 and (cap + 1) * 8 <= off < (cap + 1) * 8 + 16
        assert L.pxr_packed_rows_bytes(cap, D) == off + cap * D * 4
        sp = ops.SparseRows(cap, D, "cpu", packed=True)        # a host buffer is enough to check the aliasing
        assert sp.packed.numel() == off + cap * D * 4 and int(sp.packed.sum()) == 0
        sp.idx.fill_(-1); sp.n.fill_(5); sp.rows.fill_(1.0)
        raw = sp.packed
        assert raw[:cap * 8].view(torch.int64).eq(-1).all()
        assert int(raw[cap * 8:cap * 8 + 8].view(torch.int64)) == 5          # the count's upper half stays zero
        assert raw[cap * 8 + 8:off].eq(0).all()                               # alignment padding untouched
        assert raw[off:].view(torch.float32).eq(1.0).all()
