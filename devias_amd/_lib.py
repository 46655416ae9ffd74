"""ctypes binding of libdevias_amd.so (the C ABI declared in include/devias_amd.h).

The product path has NO fallback: if the library is missing or a symbol is absent, loading raises.
`import torch` happens first so the HIP runtime bundled with PyTorch (soname libamdhip64.so.7) is the one
the library binds to -- a second runtime is never pulled into the process.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_void_p

import torch  # noqa: F401  (must precede CDLL: see module docstring)

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DEVIAS_LIB_PATH") or os.path.join(_HERE, "libdevias_amd.so")     # (override: ablation builds of tools/)

F32, BF16 = 0, 1
ATTN_Q_PRESCALED = 1                # DEVIAS_ATTN_Q_PRESCALED (devias_mhsa_*_flags)
ABI_VERSION = 167                # devias_version() of the library these prototypes describe
ACT_NONE, ACT_GELU, ACT_RELU, ACT_SIGMOID, ACT_DGELU, ACT_DRELU = 0, 1, 2, 3, 4, 5


class GemmArgs(Structure):
    _fields_ = [
        ("A", c_void_p), ("B", c_void_p), ("C", c_void_p),
        ("M", c_int32), ("N", c_int32), ("K", c_int32),
        ("lda", c_int32), ("ldb", c_int32), ("ldc", c_int32),
        ("trans_a", c_int32), ("trans_b", c_int32),
        ("dtype", c_int32), ("c_f32", c_int32),
        ("bias", c_void_p), ("act", c_int32),
        ("aux_in", c_void_p), ("aux_out", c_void_p), ("ld_aux", c_int32),
        ("res", c_void_p), ("ldr", c_int32), ("res_mod", c_int32),
        ("beta", c_float), ("split_k", c_int32), ("ws", c_void_p),
        ("colsum", c_void_p), ("colsum_beta", c_float),
        ("row_scale", c_void_p), ("rows_per_scale", c_int32),
        ("batch", c_int32), ("stride_a", c_int64), ("stride_b", c_int64), ("stride_c", c_int64),
        ("sk_ws", c_void_p), ("sk_ws_bytes", c_int64),
    ]


class LossDims(Structure):
    _fields_ = [(n, c_int32) for n in ("B", "S", "C", "nb", "ns", "D", "G", "N", "nh")] + \
               [("w_scene", c_float), ("w_mask_pred", c_float), ("w_mask_distill", c_float), ("dtype", c_int32), ("scene_ce", c_int32)]


_FP = POINTER(c_float)


class BlockArgs(Structure):          # devias_block_args
    _fields_ = [(n, c_int32) for n in ("B", "N", "D", "H", "hidden", "dtype")] + [("eps", c_float)] + \
               [(n, c_void_p) for n in ("n1w", "n1b", "n2w", "n2b", "Wqkv", "Wp", "W1", "W2", "qkv_bias", "pb", "b1", "b2", "ds1", "ds2", "save", "ws")] + \
               [("ws_bytes", c_int64), ("sk_ws", c_void_p), ("sk_ws_bytes", c_int64)] + \
               [(n, c_void_p) for n in ("WqkvT", "WpT", "W1T", "W2T")] + \
               [("WqkvS", c_void_p), ("qkv_biasS", c_void_p)]                     # ABI 166: optional transposed weight copies for the dgrad GEMMs; 167: the q-scaled copy of Wqkv / qkv_bias


class BlockGrads(Structure):         # devias_block_grads
    _fields_ = [(n, c_void_p) for n in ("dn1w", "dn1b", "dWqkv", "dbqkv", "dWp", "dbp", "dn2w", "dn2b", "dW1", "db1", "dW2", "db2", "dx_colsum")] + \
               [("db2_done", c_int32), ("dbq", c_void_p), ("dbv", c_void_p)]


class HeadArgs(Structure):           # devias_head_args
    _fields_ = [(n, c_int32) for n in ("R", "D", "C", "h1", "h2", "G", "dtype")] + \
               [(n, c_void_p) for n in ("Wh", "W0", "W2", "W4", "bh", "b0", "b2", "b4", "ws")] + [("ws_bytes", c_int64), ("drop_mask", c_void_p)]


class HeadGrads(Structure):          # devias_head_grads
    _fields_ = [(n, c_void_p) for n in ("dWh", "dbh", "dW0", "db0", "dW2", "db2", "dW4", "db4")]


AGG_MAX_DEPTH = 16                   # DEVIAS_AGG_MAX_DEPTH
AGG_PARAM_FIELDS = ("Wq", "Wk", "Wv", "Wo", "W1", "W2", "bo", "norm_w", "norm_b", "ctx_w", "ctx_b", "b1", "b2", "ffn_w", "ffn_b")
AGG_GRAD_FIELDS = ("dWq", "dWk", "dWv", "dWo", "dbo", "dnorm_w", "dnorm_b", "dctx_w", "dctx_b", "dW1", "db1", "dW2", "db2", "dffn_w", "dffn_b")


class AggLayerParams(Structure):     # devias_agg_layer_params
    _fields_ = [(n, c_void_p) for n in AGG_PARAM_FIELDS]


class AggLayerGrads(Structure):      # devias_agg_layer_grads
    _fields_ = [(n, c_void_p) for n in AGG_GRAD_FIELDS]


class AggArgs(Structure):            # devias_agg_args
    _fields_ = [(n, c_int32) for n in ("B", "N", "S", "D", "depth", "tied", "heads", "dh", "ff", "dtype")] + \
               [("eps_enc", c_float), ("eps_agg", c_float)] + \
               [(n, c_void_p) for n in ("norm_w", "norm_b", "latents", "last_w", "last_b")] + \
               [("sets", AggLayerParams * AGG_MAX_DEPTH), ("save", c_void_p), ("ws", c_void_p), ("ws_bytes", c_int64)]


class AggGrads(Structure):           # devias_agg_grads
    _fields_ = [(n, c_void_p) for n in ("dnorm_w", "dnorm_b", "dlatents", "dlast_w", "dlast_b", "dx_colsum")] + \
               [("sets", AggLayerGrads * AGG_MAX_DEPTH)]


# name -> (restype, argtypes); mirrors include/devias_amd.h one to one
_P, _I, _L, _F = c_void_p, c_int32, c_int64, c_float
PROTOTYPES = {
    "devias_version": (c_int, []),
    "devias_last_error": (c_char_p, []),
    "devias_device_info": (c_int, [c_int, POINTER(c_int64)]),
    "devias_allreduce_bucket": (c_int, [_P, _P, _L, _I, _P]),
    "devias_shutdown": (None, []),
    "devias_counter": (c_int64, [c_int32]),
    "devias_counters_reset": (None, []),
    "devias_set_option": (c_int, [c_char_p, c_int32]),
    "devias_get_option": (c_int, [c_char_p, POINTER(c_int32)]),
    "devias_debug_mfma_probe": (c_int, [_P, _L, _I, _I, _P, _P, _P]),
    "devias_debug_mfma_probe_flops": (c_int64, [_I, _I]),
    "devias_debug_gemm_timer_arm": (c_int, [_I, _I, _I, _I, _I]),
    "devias_debug_dkdv_stamps": (c_int, [_P, _I]),
    "devias_debug_gemm_timer_read": (c_int, [POINTER(c_int32), POINTER(c_float)]),
    "devias_debug_gemm_timer_read_each": (c_int, [POINTER(c_int32), POINTER(c_float), c_int32]),
    "devias_gemm_release_queue_stream": (c_int, [_P]),
    "devias_gemm": (c_int, [POINTER(GemmArgs), _P]),
    "devias_gemm_workspace_bytes": (c_int64, [_I, _I, _I]),
    "devias_cast": (c_int, [_P, _I, _P, _I, _L, _P]),
    "devias_cast_scale": (c_int, [_P, _I, _P, _I, _L, c_float, _P]),
    "devias_patch_im2col": (c_int, [_P, _I, _P, _I, _I, _I, _I, _I, _I, _I, _I, _P]),
    "devias_colsum": (c_int, [_P, _I, _I, _I, _I, _P, _F, _P, _P]),
    "devias_colsum_workspace_bytes": (c_int64, [_I, _I]),
    "devias_rows_reduce_mod": (c_int, [_P, _I, _I, _I, _I, _P, _P]),
    "devias_rows_broadcast": (c_int, [_P, _I, _I, _P, _I, _I, _P]),
    "devias_act_bwd": (c_int, [_P, _P, _P, _I, _I, _L, _P]),
    "devias_row_scale": (c_int, [_P, _P, _I, _P, _I, _I, _I, _P]),
    "devias_add": (c_int, [_P, _P, _P, _I, _L, _P]),
    "devias_mul_mask": (c_int, [_P, _P, _P, _P, _I, _L, _P]),
    "devias_layernorm_fwd": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _F, _I, _P]),
    "devias_layernorm_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _F, _P, _I, _I, _I, _P, _P]),
    "devias_layernorm_bwd_workspace_bytes": (c_int64, [_I, _I]),
    "devias_mhsa_fwd": (c_int, [_P, _P, _P, _I, _I, _I, _F, _I, _P]),
    "devias_mhsa_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P, _P]),
    "devias_mhsa_fwd_flags": (c_int, [_P, _P, _P, _I, _I, _I, _F, _I, _I, _P]),
    "devias_mhsa_bwd_flags": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _P, _I, _P]),
    "devias_mhsa_bwd_bias_flags": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _F, ctypes.c_uint64, _P, _P, _P, _P, _I, _P]),
    "devias_mhsa_bwd_workspace_bytes": (c_int64, [_I, _I, _I]),
    "devias_mhsa_bwd_bias_dv_from_do": (c_int32, [_I, _F]),
    "devias_mhsa_fwd_dropout": (c_int, [_P, _P, _P, _I, _I, _I, _F, _I, _F, ctypes.c_uint64, _P]),
    "devias_mhsa_bwd_dropout": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _F, ctypes.c_uint64, _P]),
    "devias_mhsa_bwd_bias": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _F, _I, _F, ctypes.c_uint64, _P, _P, _P, _P, _P]),
    "devias_mhsa_bwd_bias_workspace_bytes": (c_int64, [_I, _I, _I]),
    "devias_slot_attn_fwd": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P]),
    "devias_slot_attn_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P]),
    "devias_slot_attn_kv_grad": (c_int, [_P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "devias_slot_attn_workspace_bytes": (c_int64, [_I, _I, _I, _I, _I]),
    "devias_slotf_fwd": (c_int, [_P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P]),
    "devias_slotf_bwd": (c_int, [_P, _P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _F, _I, _P, _P]),
    "devias_slotf_pack": (c_int, [_P, _P, _P, _P, _P, _P, _P, _I, _I, _I, _I, _I, _I, _I, _F, _I, _P]),
    "devias_slotf_workspace_bytes": (c_int64, [_I, _I, _I, _I, _I]),
    "devias_slot_select": (c_int, [_P, _I, _I, _I, _I, _I, _P, _P]),
    "devias_head_match_loss_fwd": (c_int, [POINTER(LossDims), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "devias_head_match_loss_bwd": (c_int, [POINTER(LossDims), _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    "devias_head_match_loss_workspace_bytes": (c_int64, [_I]),
    "devias_adamw_step": (c_int, [_P, _P, _P, _P, _L, _F, _F, _F, _F, _F, _I, _F, _P]),
    "devias_grad_sumsq_multi": (c_int, [_P, _P, _P, _I, _P, _P]),
    "devias_clip_coef": (c_int, [_P, _I, _F, _P, _P]),
    "devias_adamw_multi": (c_int, [_P, _P, _P, _I, _F, _F, _F, _F, _P, _P]),
    "devias_fame_diff_color": (c_int, [_P, _I, _I, _I, _I, _P, _P, _P]),
    "devias_fame_blur": (c_int, [_P, _P, _I, _I, _I, _I, _F, _P]),
    "devias_fame_seg_refine": (c_int, [_P, _P, _I, _I, _I, _F, _P, _P]),
    "devias_fame_binarize_pool": (c_int, [_P, _I, _I, _I, _I, _I, _P, _P, _P]),
    "devias_fame_mix": (c_int, [_P, _P, _I, _P, _P, _P, _P, _I, _I, _I, _P]),
    "devias_debug_cu_hog": (c_int, [_I, _I, _P]),
    "devias_range_push": (None, [c_char_p]),
    "devias_range_pop": (None, []),
    "devias_encoder_block_save_bytes": (c_int64, [_I, _I, _I, _I, _I, _I]),
    "devias_encoder_block_scratch_bytes": (c_int64, [_I, _I, _I, _I, _I, _I]),
    "devias_encoder_block_workspace_bytes": (c_int64, [_I, _I, _I, _I, _I, _I]),
    "devias_encoder_block_fwd": (c_int, [POINTER(BlockArgs), _P, _P, _P]),
    "devias_encoder_block_bwd": (c_int, [POINTER(BlockArgs), _P, _P, _P, POINTER(BlockGrads), _P, _L, _P]),
    "devias_head_workspace_bytes": (c_int64, [_I, _I, _I, _I, _I, _I, _I]),
    "devias_head_save_bytes": (c_int64, [_I, _I, _I, _I, _I]),
    "devias_head_fwd": (c_int, [POINTER(HeadArgs), _P, _P, _P, _P, _P]),
    "devias_head_bwd": (c_int, [POINTER(HeadArgs), _P, _P, _P, _P, _P, _P, POINTER(HeadGrads), _P]),
    "devias_agg_block_save_bytes": (c_int64, [POINTER(AggArgs)]),
    "devias_agg_block_scratch_bytes": (c_int64, [POINTER(AggArgs)]),
    "devias_agg_block_workspace_bytes": (c_int64, [POINTER(AggArgs)]),
    "devias_agg_block_fwd": (c_int, [POINTER(AggArgs), _P, _P, POINTER(c_void_p), _P]),
    "devias_agg_block_bwd": (c_int, [POINTER(AggArgs), _P, _P, _P, _P, POINTER(AggGrads), _P, _L, _P]),
    "devias_policy_gemm_cus": (c_int32, []),
    "devias_policy_small_m_split": (c_int32, [_I, _I, _I, _I]),
    "devias_policy_wgrad_split": (c_int32, [_I, _I, _I, _I]),
}
COUNTERS = {"gemm128_f32": 0, "gemm128_bf16": 1, "gemm_ss": 2, "gemm256": 3, "gemm256p": 4, "splitk_reduce": 5,
            "mhsa_fwd_bf16": 6, "mhsa_bwd_bf16": 7, "mhsa_fwd_f32": 8, "mhsa_bwd_f32": 9, "mhsa_bwd_fused": 10, "gemm_sk": 11, "gemm256w": 12, "gemm_smallm": 13, "gemm256d": 14,
            "dkdv1w": 15, "dkdv1w_pers": 16, "dkdv1w_rest": 17, "dkdv2w": 18, "mhsa_qpre": 19}     # DEVIAS_CNT_*
OPT_CHUNK = 16384          # DEVIAS_OPT_CHUNK
OPT_TENSOR_BYTES = 64      # sizeof(devias_opt_tensor)

_lib = None


class DeviasLibraryError(RuntimeError):
    pass


def load() -> ctypes.CDLL:
    """Load the shared library and bind every prototype; raises DeviasLibraryError when it is not built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise DeviasLibraryError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -m devias_amd.build` "
            "(hipcc, gfx950). There is no CPU/PyTorch fallback for the DEVIAS hot path.")
    try:
        lib = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise DeviasLibraryError(f"cannot load {LIB_PATH}: {e}") from e
    for name, (res, args) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as e:
            raise DeviasLibraryError(f"{LIB_PATH} does not export {name}; rebuild it") from e
        fn.restype = res
        fn.argtypes = args
    if lib.devias_version() < ABI_VERSION:         # an older build has a shorter devias_gemm_args: never call into it
        raise DeviasLibraryError(f"{LIB_PATH} is ABI version {lib.devias_version()}, this binding needs {ABI_VERSION}; rebuild it (python -m devias_amd.build --force)")
    _lib = lib
    return lib


CALLS = [0]                      # host -> library compute calls since the last reset (bench.py reports calls per step)


def check(rc: int, what: str) -> None:
    CALLS[0] += 1
    if rc != 0:
        msg = load().devias_last_error()
        raise RuntimeError(f"{what} failed (code {rc}): {msg.decode() if msg else '?'}")
