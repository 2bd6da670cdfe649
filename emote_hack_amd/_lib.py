"""ctypes binding of include/emo_hip.h (the C-ABI shared library built from csrc/*.hip).

There is NO fallback: if the HIP extension is missing or a call fails, this module raises.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("EMO_HIP_LIB") or os.path.join(HERE, "lib", "libemo_hip.so")

EMO_F32, EMO_BF16, EMO_F16 = 0, 1, 2


class EmoHipError(RuntimeError):
    pass


class GemmParams(C.Structure):
    _fields_ = [
        ("A", C.c_void_p), ("lda", C.c_int64),
        ("W", C.c_void_p),
        ("bias", C.c_void_p),
        ("rowbias", C.c_void_p), ("rows_per_batch", C.c_int), ("ld_rowbias", C.c_int),
        ("residual", C.c_void_p), ("ldr", C.c_int64),
        ("C", C.c_void_p), ("ldc", C.c_int64),
        ("M", C.c_int64), ("N", C.c_int), ("K", C.c_int),
        ("geglu", C.c_int),
        ("out_scale", C.c_float),
        ("transpose_out", C.c_int), ("t_rows", C.c_int), ("t_ld", C.c_int64), ("t_batch_stride", C.c_int64),
        ("conv_taps", C.c_int), ("H", C.c_int), ("W_", C.c_int), ("Cin", C.c_int), ("stride", C.c_int),
        ("upsample2x", C.c_int), ("Ho", C.c_int), ("Wo", C.c_int),
        ("dtype", C.c_int),
        ("split_k", C.c_int), ("workspace", C.c_void_p),
        ("ln_colsum", C.c_void_p), ("ln_stats", C.c_void_p),
        ("tile", C.c_int),
        ("conv_asym", C.c_int),
        ("up_h", C.c_int), ("up_w", C.c_int),
        ("w_slab_rows", C.c_int), ("w_slab_stride", C.c_int64),
        ("gn_coef", C.c_void_p), ("gn_imgs_per_inst", C.c_int), ("gn_silu", C.c_int),
        ("vt", C.c_void_p), ("vt_col0", C.c_int),
    ]


class AttentionParams(C.Structure):
    _fields_ = [
        ("q", C.c_void_p), ("ldq", C.c_int64),
        ("k0", C.c_void_p), ("ldk0", C.c_int64), ("v0t", C.c_void_p), ("ldv0t", C.c_int64), ("Lk0", C.c_int),
        ("k1", C.c_void_p), ("ldk1", C.c_int64), ("v1t", C.c_void_p), ("ldv1t", C.c_int64), ("Lk1", C.c_int),
        ("seg0_div", C.c_int),
        ("seg1_div", C.c_int), ("seg1_first_batch", C.c_int), ("seg1_skip", C.c_int),
        ("out", C.c_void_p), ("ldo", C.c_int64),
        ("B", C.c_int), ("Lq", C.c_int), ("heads", C.c_int), ("d", C.c_int),
        ("scale", C.c_float),
        ("dtype", C.c_int),
        ("seg1_row", C.c_void_p),
    ]


# name -> (restype, argtypes); every symbol include/emo_hip.h declares
_i, _i64, _f, _p, _u32 = C.c_int, C.c_int64, C.c_float, C.c_void_p, C.c_uint32
SIGNATURES = {
    "emo_version": (_i, []),
    "emo_last_error_string": (C.c_char_p, []),
    "emo_ncfhw_to_rows": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _i, _p]),
    "emo_rows_to_ncfhw": (_i, [_p, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
    "emo_copy_cols": (_i, [_p, _i, _p, _i, _i, _i64, _i, _i, _p]),
    "emo_add": (_i, [_p, _i, _p, _i, _f, _p, _i, _i64, _i, _i, _p]),
    "emo_convert": (_i, [_p, _i, _p, _i, _i64, _i, _p]),
    "emo_silu": (_i, [_p, _p, _i64, _i, _p]),
    "emo_timestep_embedding": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "emo_groupnorm_workspace_bytes": (C.c_size_t, [_i, _i64, _i, _i]),
    "emo_groupnorm_stats": (_i, [_p, _i, _p, _i, _i64, _i, _i, _i, _p]),
    "emo_groupnorm_apply": (_i, [_p, _i, _p, _p, _p, _p, _i, _i, _i64, _i, _i, _f, _i, _i, _p]),
    "emo_groupnorm_coeffs": (_i, [_p, _p, _p, _p, _i, _i64, _i, _i, _f, _i, _p]),
    "emo_groupnorm_one_launch_ok": (_i, [_i, _i64, _i, _i, _i]),
    "emo_groupnorm": (_i, [_p, _i, _p, _p, _p, _i, _i, _i64, _i, _i, _f, _i, _i, _p]),
    "emo_groupnorm_fold_linear": (_i, [_p, _p, _p, _p, _p, _p, _p, _i, _i64, _i, _i, _i, _f, _i, _p]),
    "emo_layernorm": (_i, [_p, _i, _p, _p, _p, _i, _i64, _i, _f, _p, _i, _i, _i, _p]),
    "emo_layernorm_stats": (_i, [_p, _i, _p, _i64, _i, _f, _i, _p]),
    "emo_gemm": (_i, [C.POINTER(GemmParams), _p]),
    "emo_conv3x3_gn_fusable": (_i, [C.POINTER(GemmParams)]),
    "emo_gemm_vt_ok": (_i, [C.POINTER(GemmParams)]),
    "emo_gemm_suggest_split_k": (_i, [_i64, _i, _i, _i, _i, _i]),
    "emo_gemm_workspace_bytes": (C.c_size_t, [_i64, _i, _i]),
    "emo_attention": (_i, [C.POINTER(AttentionParams), _p]),
    "emo_temporal_attention": (_i, [_p, _i64, _p, _i64, _i, _i, _i, _i, _i, _f, _i, _p]),
    "emo_cfg_step": (_i, [_p, _p, _p, _p, _i, _i, _i, _f, _f, _f, _f, _u32, _u32, _p]),
    "emo_accumulate_window": (_i, [_p, _i, _p, _p, _p, _i, _i, _i, _i, _i, _i, _p]),
    "emo_act": (_i, [_p, _p, _i64, _i, _i, _p]),
    "emo_speed_encode": (_i, [_p, _p, _p, _p, _i, _i, _i, _p]),
    "emo_speed_bucket": (_i, [_p, _p, _p, _i, _i, _p]),
    "emo_gather_rows": (_i, [_p, _p, _p, _i, _i, _i, _i, _p]),
    "emo_add_rowbias": (_i, [_p, _i, _p, _i, _p, _i, _i64, _i, _i, _i, _p]),
    "emo_softmax_rows": (_i, [_p, _i64, _p, _i64, _i64, _i, _f, _i, _p]),
    "emo_audio_windows": (_i, [_p, _p, _i, _i, _i, _i, _i, _p]),
    "emo_rows_to_video": (_i, [_p, _i64, _p, _i, _i, _i, _i, _f, _f, _f, _f, _i, _p]),
    "emo_channelnorm_workspace_bytes": (C.c_size_t, [_i64, _i]),
    "emo_channelnorm": (_i, [_p, _i64, _p, _p, _p, _i64, _i64, _i, _f, _i, _p, _i, _p]),
    "emo_maxpool2x2": (_i, [_p, _i64, _p, _i64, _i, _i, _i, _i, _i, _p]),
    "emo_bilinear_to_nchw": (_i, [_p, _i64, _p, _i, _i, _i, _i, _i, _i, _i, _p]),
}

_lib = None


def load():
    """Load libemo_hip.so (built by emote_hack_amd.build.build_extension / __graft_entry__.build)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise EmoHipError(
            f"HIP extension not built: {LIB_PATH} is missing. Run `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback for the product path.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is not exported
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def check(rc: int, what: str = ""):
    if rc != 0:
        msg = load().emo_last_error_string()
        raise EmoHipError(f"{what} failed (emo_status {rc}): {msg.decode() if msg else ''}")
