"""ctypes binding of libstemgnn_hip.so (C ABI declared in include/stemgnn_hip.h).

Fails loudly: a missing / unloadable library raises at first use -- there is no fallback path.
"""
import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int, c_long, c_size_t, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("STEMGNN_HIP_LIB", os.path.join(_HERE, "libstemgnn_hip.so"))   # override: A/B builds only
SG_BLOCK_NPARAMS = 33
SG_EINVAL = -10001

_P = c_void_p          # device pointer
_PP = POINTER(c_void_p)  # host array of device pointers

# name -> (restype, argtypes); mirrors include/stemgnn_hip.h one to one
SIGNATURES = {
    "stemgnn_version": (c_char_p, []),
    "stemgnn_num_cus": (c_int, []),
    "stemgnn_table_floats": (c_size_t, [c_int, c_int]),
    "stemgnn_packed_floats": (c_size_t, [c_int, c_int]),
    "stemgnn_saved_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "stemgnn_scratch_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "stemgnn_scratch_offset_dG": (c_size_t, [c_int, c_int, c_int, c_int]),
    "stemgnn_gradpart_floats": (c_size_t, [c_int, c_int, c_int]),
    "stemgnn_attn_saved_floats": (c_size_t, [c_int, c_int]),
    "stemgnn_attn_scratch_floats": (c_size_t, [c_int, c_int, c_int]),
    "stemgnn_make_tables_host": (c_int, [c_int, c_int, _P]),
    "stemgnn_attn_laplacian_fwd": (c_int, [_P, _P, _P, c_float, c_float, c_int, _P, c_int, c_int, _P, _P, _P, c_int, _P]),
    "stemgnn_attn_laplacian_bwd": (c_int, [_P, _P, _P, _P, c_float, c_float, c_int, _P, c_int, c_int, _P, _P, c_int,
                                           _P, _P, _P, c_int, _P]),
    "stemgnn_dropout_mask": (c_int, [c_float, _P, c_int, c_int, _P, _P]),
    "stemgnn_dropout_seed_next": (c_int, [_P, _P, _P]),
    "stemgnn_cheb_fwd": (c_int, [_P, c_int, _P]),
    "stemgnn_cheb_bwd": (c_int, [_P, _P, _P, _P, c_int, _P]),
    "stemgnn_eigh_scratch_floats": (c_size_t, [c_int]),
    "stemgnn_eigh_fwd": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "stemgnn_eigh_batched": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "stemgnn_eigh_status": (c_int, []),
    "stemgnn_eigh_cluster_fixes": (c_int, []),
    "stemgnn_split_planes_floats": (c_size_t, [c_int, c_int, c_int]),
    "stemgnn_split_weights_bf16": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "stemgnn_glu_gemm_bf16": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "stemgnn_glu_gemm_f32": (c_int, [_P, _P, _P, c_int, c_int, c_int, _P]),
    "stemgnn_sgemm_f32": (c_int, [_P, c_int, c_int, _P, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "stemgnn_glu_combine_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, c_int, c_int, _P]),
    "stemgnn_glu_combine_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P]),
    "stemgnn_colsum": (c_int, [_P, c_int, c_int, _P, _P]),
    "stemgnn_glu_fused_repack": (c_int, [_P, c_int, c_int, _P]),
    "stemgnn_shortcut_dx": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "stemgnn_gru_reserve_floats": (c_size_t, [c_int, c_int, c_int]),
    "stemgnn_gru_fwd_scratch_floats": (c_size_t, [c_int, c_int, c_int]),
    "stemgnn_gru_bwd_scratch_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "stemgnn_gru_bwd_cus": (c_int, [c_int, c_int]),
    "stemgnn_gru_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P]),
    "stemgnn_gru_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "stemgnn_gru_bwd_rank2_ok": (c_int, [c_int, c_int]),
    "stemgnn_gru_bwd_rank2": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "stemgnn_gru_bwd_rank2_dq": (c_int, [_P, _P, c_int, c_int, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "stemgnn_gru_bwd_overlap_ok": (c_int, [c_int, c_int, c_int, c_int]),
    "stemgnn_gru_bwd_ctl_words": (c_size_t, [c_int]),
    "stemgnn_gru_bwd_rank2_begin": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P]),
    "stemgnn_gru_bwd_rank2_finish": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "stemgnn_keyquery_wgrad": (c_int, [_P, _P, _P, _P, c_int, c_int, _P]),
    "stemgnn_attn_dquery_reduce": (c_int, [_P, c_int, c_int, c_int, _P, _P]),
    "stemgnn_keyquery_wgrad2": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, _P]),
    "stemgnn_fc_tail_supported": (c_int, [c_int, c_int]),
    "stemgnn_fc_tail_scratch_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "stemgnn_fc_tail_fwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P]),
    "stemgnn_fc_tail_bwd": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "stemgnn_fill_zero": (c_int, [_P, c_size_t, _P]),
    "stemgnn_rmsprop_step": (c_int, [_P, _P, _P, c_size_t, _P, c_float, c_float, c_int, c_float, _P]),
    "stemgnn_adam_step": (c_int, [_P, _P, _P, _P, c_size_t, _P, _P, c_float, c_float, c_float, c_int, c_float, _P]),
    "stemgnn_normalize_series": (c_int, [_P, _P, _P, c_int, _P, c_long, c_int, _P]),
    "stemgnn_window_gather": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_long, _P, _P]),
    "stemgnn_window_gather_queue": (c_int, [_P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, c_long, _P, _P]),
    "stemgnn_mse_scratch_floats": (c_size_t, []),
    "stemgnn_mse_fwd": (c_int, [_P, _P, c_size_t, _P, _P, _P, _P]),
    "stemgnn_mse_bwd": (c_int, [_P, _P, c_size_t, _P, _P, _P]),
    "stemgnn_roll_window": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "stemgnn_eval_scratch_doubles": (c_size_t, [c_long, c_int, c_int]),
    "stemgnn_eval_out_doubles": (c_size_t, [c_int, c_int]),
    "stemgnn_eval_metrics": (c_int, [_P, _P, _P, _P, c_long, c_int, c_int, _P, _P, _P]),
    "stemgnn_block_pack": (c_int, [_PP, _P, _P, c_int, c_int, _P]),
    "stemgnn_block_pack_panels": (c_int, [_PP, _P, _P, c_int, c_int, _P]),
    "stemgnn_block_unpack_grads": (c_int, [_P, c_int, _P, _PP, c_int, c_int, c_int, _P]),
    "stemgnn_gft_fwd": (c_int, [_P, _P, c_long, c_long, c_long, _P, c_int, c_int, c_int, _P]),
    "stemgnn_gft_bwd": (c_int, [_P, _P, c_long, c_long, c_long, _P, _P, _P, c_int, c_int, c_int, c_int, _P]),
    "stemgnn_gft_bwd_dt2": (c_int, [_P, c_long, c_long, c_long, _P, _P, c_long, c_long, c_long, _P, _P, c_int, c_int, c_int, _P]),
    "stemgnn_spectral_glu_fwd": (c_int, [_P, _P, c_int, c_int, c_int, c_int, _P]),
    "stemgnn_spectral_glu_bwd": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P]),
    "stemgnn_glu_split_floats": (c_size_t, [c_int, c_int, c_int]),
    "stemgnn_glu_split_panels": (c_int, [_P, _P, c_int, c_int, c_int, _P]),
    "stemgnn_glu_fused_bf16_ok": (c_int, [c_int, c_int, c_int]),
    "stemgnn_spectral_glu_fwd_split": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "stemgnn_glu_warm_saved_floats": (c_size_t, [c_int, c_int]),
    "stemgnn_spectral_glu_fwd_warm": (c_int, [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "stemgnn_spectral_glu_dgrad_split": (c_int, [_P, _P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P]),
    "stemgnn_igft_heads_fwd": (c_int, [_PP, _P, _P, _P, c_long, c_long, c_long, _P, c_int, _P,
                                       c_int, c_int, c_int, c_int, _P]),
    "stemgnn_igft_heads_bwd": (c_int, [_PP, _P, _P, _P, c_long, c_long, c_long, _P, _P, _P, _P, _P, c_int, c_int,
                                       c_int, c_int, c_int, c_int, _P]),
    "stemgnn_fc_tail_train_scratch_floats": (c_size_t, [c_int, c_int, c_int, c_int]),
    "stemgnn_fc_tail_train": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P,
                                      _P, _P]),
    "stemgnn_fc_tail_train_rows": (c_int, [_P, _P, _P, _P, _P, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P]),
    "stemgnn_fc_tail_train_finish": (c_int, [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P]),
    "stemgnn_block_wgrad": (c_int, [_PP, _P, _P, _P, c_long, c_long, c_long, _P, c_int, _P, _P, c_int, c_int,
                                    c_int, c_int, c_int, c_int, _P]),
    "stemgnn_block_wgrad_split": (c_int, [_PP, _P, _P, _P, c_long, c_long, c_long, _P, c_int, _P, _P, c_int, c_int,
                                          c_int, c_int, c_int, c_int, c_int, _P]),
}

_lib = None


class StemGNNHipError(RuntimeError):
    pass


def load():
    """Load (once) and return the ctypes handle; raises if the HIP library is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.isfile(LIB_PATH):
        raise StemGNNHipError(
            f"{LIB_PATH} not found -- build it with `python __graft_entry__.py` (hipcc --offload-arch=gfx950); "
            "stemgnn_amd has no CPU / eager fallback.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(rc, what):
    if rc != 0:
        if rc == SG_EINVAL:
            raise StemGNNHipError(f"{what}: invalid argument (SG_EINVAL)")
        raise StemGNNHipError(f"{what}: HIP error {-rc}")


def ptr_array(tensors):
    """host array of SG_BLOCK_NPARAMS device pointers (None -> NULL)."""
    arr = (c_void_p * SG_BLOCK_NPARAMS)()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr
