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
    # one ABI revision in three places: the header, the ctypes binding, the built library.  0.3.0 = the revision in which the
    # `stat` buffers of pxr_ln_bwd_stat_f32 / pxr_attn_bwd_stat_f32 changed size under unchanged names
    hdr = int(re.search(r"#define\s+PXR_ABI_VERSION\s+(\d+)", open(os.path.join(ROOT, "include", "pxr.h")).read()).group(1))
    assert hdr == lib.ABI_VERSION == L.pxr_version() == 300
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


# ---- round 5: ISA gate ---------------------------------------------------------------------------
# Scratch (register spills / dynamically indexed local arrays) of every kernel in the built library, from the AMDGPU metadata of
# the gfx950 code objects inside the fat binary (tools/isa_resources.py: .private_segment_fixed_size).  No GPU needed.  Since
# round 5 every gemm_p3_kernel instantiation -- the GEMM behind every nn.Linear of the step and of the image tower -- is scratch
# free (rounds 3-4: 48 .. 272 B per lane in the epilogues: operands requested before the main loop were parked in scratch across
# it, and the epilogue lambda was not inlined, so its arrays lived in memory).  What is left is listed with its budget: a kernel
# that starts to spill, or one of these that spills more, fails here.
SCRATCH_BUDGET = [
    (r"grouped_dw_p3_kernel<.*, 2>\(", 16),              # split-K weight gradients: 8 / 12 B in the store-then-add epilogue
    (r"tower_attn_bwd_dkv_kernel<[5-8]>", 84),           # ViT-B/16 tower attention backward (one 7-wave workgroup per CU)
    (r"tower_attn_bwd_dkv_kernel<9>", 432), (r"tower_attn_bwd_dq_kernel<9>", 236), (r"tower_attn_fwd_kernel<9, false>", 12),   # 257-token towers
    (r"score_thresh_p3_kernel", 16),                     # six-product top-k pass
    (r"score_topk2_kernel<", 600), (r"score_thresh_kernel<[01]>", 128),       # fallback top-k variants (PXR_TOPK_VARIANT / no planes)
]


def test_isa_gate_no_kernel_spills_outside_the_listed_budgets():
    import re
    import sys

    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import isa_resources

    res = isa_resources.kernel_resources()
    assert len(res) > 500                                                   # the whole library was read (819 kernels in round 5)
    gemms = {n: r for n, r in res.items() if "gemm_p3_kernel<" in n}
    assert len(gemms) > 100 and all(r["scratch"] == 0 for r in gemms.values()), {n: r["scratch"] for n, r in gemms.items() if r["scratch"]}
    # the kernels of the headline step (profiles/r05/bench_b64_step_timeline.txt) by name: present and scratch free
    for hot in (r"gemm_p3_kernel<pxr::P3Cfg<128, 64, 2, 2, 2, true>, true, 1, false>", r"gemm_p3_kernel<pxr::P3Cfg<128, 64, 2, 2, 2, true>, true, 5, false>",
                r"gemm_p3_kernel<pxr::P3Cfg<64, 64, 2, 2, 3, true>, false, 4, false>", r"gemm_p3_kernel<pxr::P3Cfg<128, 64, 2, 2, 2, true>, false, 6, false>",
                r"attn_fwd_mfma1_kernel<8>", r"attn_bwd_mfma1_kernel<8>", r"ln_fwd_kernel<2, true, 1, false>", r"ln_fwd_kernel<2, false, 1, false>", r"ln_fwd_kernel<2, true, 1, true>",
                r"ln_fwd_kernel<2, false, 1, true>", r"ln_fwd_kernel<4, false, 1, true>",
                r"ln_bwd_kernel<2, false, false>", r"ln_bwd_kernel<2, true, false>", r"ln_bwd_kernel<2, false, true>", r"adamw_rows_kernel<256, 2>", r"adamw_flat_tab_kernel<true>",
                r"segsum_kernel<1>", r"fused_pass_kernel<1>", r"h2_split_auto_kernel", r"score_thresh_fast_kernel<2, 4>", r"topk_rescore_kernel"):
        hits = [n for n in res if re.search(re.escape(hot), n)]
        assert hits, hot
        assert all(res[n]["scratch"] == 0 for n in hits), (hot, [res[n]["scratch"] for n in hits])
    over = {}
    for name, r in res.items():
        if r["scratch"] == 0:
            continue
        budget = max((b for pat, b in SCRATCH_BUDGET if re.search(pat, name)), default=0)
        if r["scratch"] > budget:
            over[name] = (r["scratch"], budget)
    assert not over, over
