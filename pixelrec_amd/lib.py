"""ctypes binding of libpxr.so -- the C-ABI library of hand-written gfx950 HIP kernels.

The product path has NO CPU fallback: if the library is missing or a symbol is absent this module raises,
loudly, instead of routing around it (the CPU restatement under ``oracle/`` is test infrastructure only and is
never imported from here).

PyTorch is used for device memory and streams only: every wrapper hands raw ``data_ptr()`` addresses and the
current HIP stream to the C entry points declared in ``include/pxr.h``.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_int64, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpxr.so")


class PxrError(RuntimeError):
    pass


_lib = None

# name -> (restype, argtypes).  Must list every symbol include/pxr.h declares (tests/test_abi.py checks both
# directions against the header).
_P = c_void_p
_F, _D, _I, _I64, _U64, _U32 = c_float, c_double, c_int, c_int64, c_uint64, ctypes.c_uint32
_SIGNATURES = {
    "pxr_version": (c_int, []),
    "pxr_last_error": (c_char_p, []),
    "pxr_target_arch": (c_char_p, []),
    "pxr_dropout_keep_host": (_I, [_U64, _U32, _U64, _I64, _F, _P]),
    "pxr_embed_gather_f32": (_I, [_P, _I64, _I, _P, _I64, _P, _P]),
    "pxr_embed_grad_ws_bytes": (_I64, [_I64]),
    "pxr_embed_grad_rows_f32": (_I, [_P, _I64, _P, _I, _I64, _F, _P, _P, _P, _P, _I64, _P]),
    "pxr_sample_negatives_i64": (_I, [_P, _I, _I, _I64, _U64, _U64, _P, _P, _P]),
    "pxr_shard_local_rows_i64": (_I, [_P, _I64, _I, _I, _I64, _P, _P]),
    "pxr_shard_bucket_ids_i64": (_I, [_P, _P, _I, _I64, _I64, _I64, _P, _P, _P, _P]),
    "pxr_scatter_rows_f32": (_I, [_P, _P, _I64, _I, _P, _I64, _I, _P]),
    "pxr_ids_to_compact_i64": (_I, [_P, _I64, _P, _P, _P, _P]),
    "pxr_shard_first_rows_i64": (_I, [_P, _I, _I64, _I, _I64, _P, _P]),
    "pxr_merge_rows_ws_bytes": (_I64, [_I, _I64]),
    "pxr_merge_sorted_rows_f32": (_I, [_P, _P, _I, _I64, _I, _I64, _F, _P, _P, _P, _P, _I64, _P]),
    "pxr_packed_rows_offset": (_I64, [_I64]),
    "pxr_packed_rows_bytes": (_I64, [_I64, _I]),
    "pxr_merge_packed_rows_f32": (_I, [_P, _I, _I64, _I, _I64, _F, _P, _P, _P, _P, _I64, _P]),
    "pxr_sasrec_embed_grad_f32": (_I, [_P, _I, _I, _P, _P, _P, _I, _I64, _F, _P, _P, _P, _P, _I64, _P]),
    "pxr_sasrec_occ_sort": (_I, [_P, _I, _I, _I64, _P, _P, _P, _I64, _P]),
    "pxr_sasrec_occ_segsum": (_I, [_P, _I64, _I, _I, _P, _P, _P, _I, _I64, _F, _P, _P, _P]),
    "pxr_sasrec_occ_split_ws_bytes": (_I64, [_I, _I, _I]),
    "pxr_sasrec_occ_segsum_split": (_I, [_P, _I64, _I, _I, _P, _P, _P, _I, _I64, _F, _P, _P, _P, _I64, _P]),
    "pxr_input_ln_fwd_f32": (_I, [_P, _I64, _P, _I64, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _F, _U64, _U32, _P, _P]),
    "pxr_ln_residual_fwd_f32": (_I, [_P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _F, _U64, _U32, _P, _P]),
    "pxr_ln_bwd_ws_bytes": (_I64, [_I, _I]),
    "pxr_ln_bwd_partial_rows": (_I, [_I]),
    "pxr_colsum_partial_rows": (_I, [_I]),
    "pxr_reduce_partials_multi_f32": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pxr_ln_bwd_f32": (_I, [_I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _F, _U64, _U32, _P, _P, _I64, _P]),
    "pxr_gemm_ws_bytes": (_I64, [_I, _I, _I, _I, _I]),
    "pxr_gemm_f32": (_I, [_I, _I, _I, _I, _I, _P, _I64, _P, _I64, _P, _I64, _I, _P, _P, _I64, _P, _I64, _I, _I, _P]),
    "pxr_linear_fwd_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _P]),
    "pxr_linear_bwd_input_f32": (_I, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _P]),
    "pxr_linear_bwd_weight_f32": (_I, [_P, _P, _P, _I, _I, _I, _P, _I64, _P]),
    "pxr_grouped_linear_bwd_weight_f32": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pxr_colsum_ws_bytes": (_I64, [_I, _I]),
    "pxr_colsum_f32": (_I, [_P, _I64, _I, _I, _P, _P, _I64, _P]),
    "pxr_attn_fwd_f32": (_I, [_P, _P, _P, _I64, _P, _I64, _I, _I, _I, _I, _P, _I64, _P, _F, _U64, _U32, _P, _P]),
    "pxr_attn_bwd_f32": (_I, [_P, _I64, _P, _P, _P, _I64, _P, _I, _I, _I, _I, _P, _P, _P, _I64, _F, _U64, _U32, _P, _P]),
    "pxr_bpr_loss_fwd_f32": (_I, [_P, _P, _I64, _P, _P, _I, _I, _I, _P, _P, _P, _P, _P]),
    "pxr_bpr_loss_bwd_f32": (_I, [_P, _P, _P, _I64, _P, _P, _I, _I, _I, _F, _P, _P, _P, _P]),
    "pxr_bpr_loss_reduce_f32": (_I, [_P, _I, _I, _P, _P]),
    "pxr_ln_residual_bpr_fwd_f32": (_I, [_P, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _F, _U64, _U32, _P, _P, _I64, _P, _P, _P, _P, _P,
                                         _P, _P]),
    "pxr_bpr_ln_bwd_f32": (_I, [_P, _P, _P, _I64, _P, _P, _I, _I, _F, _P, _P, _P, _P, _P, _I, _P, _P, _P, _P, _F, _U64, _U32, _P, _P,
                                _I64, _P, _I64, _I64, _P, _P]),
    "pxr_mosasrec_emb_grad_f32": (_I, [_P, _P, _P, _I, _I, _I, _P, _P]),
    "pxr_image_u8_to_f32": (_I, [_P, _I64, _I, _I, _P, _I, _P, _P]),
    "pxr_score_topk_ws_bytes": (_I64, [_I, _I, _I]),
    "pxr_score_topk_f32": (_I, [_P, _I64, _I, _P, _I, _I, _P, _P, _I, _P, _P, _P, _I64, _P]),
    "pxr_score_topk_planes_f32": (_I, [_P, _I64, _I, _P, _I, _I, _P, _I64, _I64, _P, _I64, _I64, _P, _P, _I, _P, _P, _P, _I64, _P]),
    "pxr_score_topk_fast_f32": (_I, [_P, _I64, _I, _P, _I, _I, _P, _I64, _I64, _P, _I64, _I64, _P, _I, _P, _P, _I, _P, _P, _P, _I64, _P]),
    "pxr_row_norm_max_f32": (_I, [_P, _I64, _I64, _I64, _P, _P]),
    "pxr_adamw_flat_f32": (_I, [_P, _P, _P, _P, _I64, _D, _D, _D, _D, _D, _I64, _P]),
    "pxr_slot_fill_i32": (_I, [_P, _I64, ctypes.c_int32, _P]),
    "pxr_adamw_table_f32": (_I, [_P, _P, _P, _I64, _I, _P, _P, _P, _P, _I64, _D, _D, _D, _D, _D, _I64, _P]),
    "pxr_adamw_hyper_append": (_I, [_P, _P, _I64, _I64, _P, _D, _D, _D, _D, _D, _I, _P]),
    "pxr_adamw_rows_f32": (_I, [_P, _P, _P, _P, _I64, _I, _P, _P, _I64, _P, _P, _P, _I64, _I64, _P, _I64, _I64, _D, _D, _D, _P]),
    "pxr_gemm_reset_flags": (_I, []),
    "pxr_score_topk_clock_out": (_I, [_P]),
    "pxr_adamw_rows_ids_f32": (_I, [_P, _P, _P, _P, _I64, _I, _P, _I64, _P, _P, _I64, _P, _D, _D, _D, _P]),
    "pxr_adamw_rows_ids2d_f32": (_I, [_P, _P, _P, _P, _I64, _I, _P, _I64, _I64, _I64, _P, _P, _I64, _P, _D, _D, _D, _P, _P]),
    "pxr_adamw_flat_tab_ex_f32": (_I, [_P, _P, _P, _P, _I64, _P, _P, _I64, _I64, _P, _P, _D, _D, _D, _D, _D, _I, _P, _P, _P, _P, _P, _P,
                                       _I, _P, _P]),
    "pxr_adamw_flat_tab_f32": (_I, [_P, _P, _P, _P, _I64, _P, _I64, _P, _D, _D, _D, _P]),
    "pxr_adamw_flat_tab_planes_f32": (_I, [_P, _P, _P, _P, _I64, _P, _I64, _P, _D, _D, _D, _I, _P, _P, _P, _P, _P, _P, _P]),
    "pxr_counter_add_i64": (_I, [_P, _I64, _P]),
    "pxr_set_status_word": (_I, [_P]),
    "pxr_gemm_batched_f32": (_I, [_I, _I, _I, _I, _I, _P, _I64, _P, _I64, _P, _I64, _I, _I, _I64, _I64, _I64, _I64, _I64,
                                  _I64, _I, _P]),
    "pxr_set_gemm_mode": (_I, [_I]),
    "pxr_get_gemm_mode": (_I, []),
    "pxr_tower_attn_supported": (_I, [_I, _I]),
    "pxr_tower_attn_fwd_f32": (_I, [_P, _P, _P, _I64, _I64, _I, _I, _I, _F, _P, _I64, _P, _I64, _I64, _P, _P]),
    "pxr_tower_attn_bwd_f32": (_I, [_P, _P, _P, _I64, _P, _P, _I64, _P, _I64, _I, _I, _I, _F, _P, _P, _P, _I64, _P, _P]),
    "pxr_causal_im2col_f32": (_I, [_P, _P, _I64, _I, _I, _I, _I, _P]),
    "pxr_causal_col2im_f32": (_I, [_P, _P, _I64, _I, _I, _I, _I, _P]),
    "pxr_gru_gates_fwd_f32": (_I, [_P, _P, _P, _P, _P, _I64, _I, _P]),
    "pxr_gru_gates_bwd_f32": (_I, [_P, _P, _P, _P, _P, _P, _I64, _I, _P]),
    "pxr_softmax_rows_f32": (_I, [_P, _I64, _I, _I, _F, _P]),
    "pxr_softmax_rows_bwd_f32": (_I, [_P, _P, _I64, _I, _I, _F, _P]),
    "pxr_vit_embed_f32": (_I, [_P, _P, _P, _P, _I64, _I, _I, _P]),
    "pxr_token_mean_f32": (_I, [_P, _P, _I64, _I, _I, _P]),
    "pxr_token_mean_relu_bwd_f32": (_I, [_P, _P, _P, _I64, _I, _I, _P]),
    "pxr_add_f32": (_I, [_P, _P, _P, _I64, _P]),
    "pxr_dropout_f32": (_I, [_P, _P, _I64, _F, _U64, _U32, _P, _P]),
    "pxr_attn_rows_fwd_f32": (_I, [_P, _P, _P, _I64, _I, _I, _I, _I, _F, _U64, _U32, _P, _I, _P]),
    "pxr_attn_rows_bwd_f32": (_I, [_P, _P, _I, _I, _I, _I, _F, _U64, _U32, _P, _I, _P]),
    "pxr_split_planes_f32": (_I, [_P, _I64, _I64, _I64, _P, _I64, _I64, _P]),
    "pxr_gemm_planes_f32": (_I, [_I, _I, _I, _I, _P, _I64, _I64, _P, _I64, _I64, _P, _I64, _I, _P, _P, _I64, _P, _I64, _I64,
                                 _I, _I, _P]),
    "pxr_split_planes_multi_f32": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pxr_split_h2_multi_f32": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "pxr_gemm_h2_f32": (_I, [_I, _I, _I, _I, _P, _I64, _I64, _I, _P, _P, _I64, _I64, _I, _P, _P, _I64, _I, _P, _P, _I64, _P, _I64, _I64,
                             _I, _P, _I, _I, _P]),
    "pxr_grouped_dw_h2_f32": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "pxr_h2_split_auto_multi_f32": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _I, _P, _P, _P]),
    "pxr_h2_bound_exp": (_I, [_P, _P, _F, _P, _P]),
    "pxr_ln_bwd_stat_f32": (_I, [_P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _F, _U64, _U32, _P, _P, _I64, _P, _P, _I, _P]),
    "pxr_ln_bwd_res_f32": (_I, [_P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _I64, _P, _P]),
    "pxr_h2_split_parts_f32": (_I, [_P, _I64, _I64, _I64, _P, _I64, _I64, _P, _I, _P, _P, _P, _F, _P, _P]),
    "pxr_ln_bwd_h2s_f32": (_I, [_P, _P, _P, _I64, _P, _P, _I, _I, _F, _P, _P, _P, _P, _P, _P, _I, _I, _P, _P, _P, _F, _U64, _U32, _P, _P,
                                _I64, _P, _I64, _I64, _P, _P, _P, _I, _P]),
    "pxr_attn_bwd_h2s_f32": (_I, [_P, _I64, _P, _P, _P, _I64, _P, _I, _I, _I, _I, _F, _U64, _U32, _P, _P, _I64, _I64, _I, _I, _I, _I, _P,
                                  _P, _P]),
    "pxr_h2_sites_update": (_I, [_I, _P, _P, _P, _P, _F, _I, _F, _P, _P, _P, _P, _P]),
    "pxr_attn_bwd_stat_f32": (_I, [_P, _I64, _P, _P, _P, _I64, _P, _I, _I, _I, _I, _P, _P, _P, _I64, _F, _U64, _U32, _P, _P, _P]),
    "pxr_ln_residual_fwd_h2_f32": (_I, [_P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _F, _U64, _U32, _P, _P, _I64, _I64, _P]),
    "pxr_input_ln_fwd_h2_f32": (_I, [_P, _I64, _P, _I64, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _F, _U64, _U32, _P, _P, _I64, _I64, _P]),
    "pxr_attn_fwd_h2_f32": (_I, [_P, _P, _P, _I64, _P, _I64, _I, _I, _I, _I, _P, _I64, _P, _F, _U64, _U32, _P, _P, _I64, _I64, _P]),
    "pxr_tower_attn_fwd_h2_f32": (_I, [_P, _P, _P, _I64, _I64, _I, _I, _I, _F, _P, _I64, _P, _I64, _I64, _P, _P]),
    "pxr_input_ln_fwd_planes_f32": (_I, [_P, _I64, _P, _I64, _P, _P, _P, _F, _I, _I, _I, _P, _P, _P, _F, _U64, _U32, _P, _P, _I64,
                                         _I64, _P]),
    "pxr_ln_residual_fwd_planes_f32": (_I, [_P, _P, _P, _P, _F, _I, _I, _P, _P, _P, _F, _U64, _U32, _P, _P, _I64, _I64, _P]),
    "pxr_ln_bwd_planes_f32": (_I, [_I, _P, _P, _P, _P, _I, _I, _P, _P, _P, _P, _F, _U64, _U32, _P, _P, _I64, _P, _I64, _I64, _P]),
    "pxr_attn_planes_supported": (_I, [_I, _I]),
    "pxr_attn_fwd_planes_f32": (_I, [_P, _P, _P, _I64, _P, _I64, _I, _I, _I, _I, _P, _I64, _P, _F, _U64, _U32, _P, _P, _I64, _I64,
                                     _P]),
    "pxr_attn_bwd_planes_f32": (_I, [_P, _I64, _P, _P, _P, _I64, _P, _I, _I, _I, _I, _P, _P, _P, _I64, _F, _U64, _U32, _P, _P, _I64,
                                     _I64, _I, _I, _I, _I, _P]),
    "pxr_grouped_dw_planes_f32": (_I, [_I, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _P]),
    "pxr_merge_split_rows_f32": (_I, [_P, _P, _I, _I64, _I64, _I, _I64, _F, _P, _P, _P, _P, _I64, _P]),
}


ABI_VERSION = 300      # include/pxr.h PXR_ABI_VERSION (tests/test_abi.py pins header == binding == library)


def load():
    """Load libpxr.so (once) and bind every declared entry point.  Raises PxrError if anything is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise PxrError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(or `make -C pixelrec_amd/csrc`). pixelrec_amd has no CPU fallback."
        )
    # torch bundles its own libamdhip64.so.7; it must be the HIP runtime instance in this process (device memory and
    # streams come from torch), so torch is imported BEFORE libpxr.so resolves the same soname.
    import torch  # noqa: F401

    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover - depends on the host's ROCm install
        raise PxrError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in _SIGNATURES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise PxrError(f"{LIB_PATH} does not export `{name}` (stale build?)") from e
        fn.restype = res
        fn.argtypes = args
    if lib.pxr_version() != ABI_VERSION:
        raise PxrError(f"{LIB_PATH} reports ABI revision {lib.pxr_version()}, this binding was written for {ABI_VERSION} "
                       "(include/pxr.h PXR_ABI_VERSION): rebuild the library -- entries change meaning between revisions")
    _lib = lib
    return lib


def exported_symbols():
    return sorted(_SIGNATURES)


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().pxr_last_error()
        raise PxrError(f"{what or 'pxr call'} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Raw device address of a (contiguous or deliberately strided) torch tensor, or None."""
    if t is None:
        return None
    return c_void_p(t.data_ptr())


_raw_stream = None
_dev_index = None


def raw_stream() -> int:
    """Raw hipStream_t of torch's CURRENT stream on this process' device (one process per GPU).  Uses the same
    fast C accessor as torch's compiled-kernel launchers; `torch.cuda.current_stream()` costs ~10 us per call."""
    global _raw_stream, _dev_index
    import torch

    if _raw_stream is None:
        _raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None) or False
    if _dev_index is None:
        _dev_index = torch.cuda.current_device()
    if _raw_stream:
        return _raw_stream(_dev_index)
    return torch.cuda.current_stream().cuda_stream


def stream_ptr():
    return c_void_p(raw_stream())
