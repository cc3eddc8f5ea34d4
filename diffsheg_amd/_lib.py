"""ctypes binding of libdiffsheg_hip.so (include/diffsheg_hip.h).

There is deliberately no fallback: if the HIP library is missing the import of any product module
fails with a clear message (build it with ``python -c 'import __graft_entry__ as g; g.build()'`` or
``make -C diffsheg_amd/csrc``).
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libdiffsheg_hip.so")


class DshError(RuntimeError):
    pass


class ModelConfigC(C.Structure):
    _fields_ = [("dim_pose", C.c_int32), ("expression_dim", C.c_int32), ("style_dim", C.c_int32),
                ("classifier_free", C.c_int32), ("cond_scale", C.c_float), ("latent_dim", C.c_int32),
                ("ff_size", C.c_int32), ("num_layers", C.c_int32), ("num_heads", C.c_int32),
                ("audio_dim", C.c_int32), ("aud_latent_dim", C.c_int32), ("hubert_dim", C.c_int32),
                ("hubert_enc_dim", C.c_int32), ("precision", C.c_int32), ("single_transformer", C.c_int32)]


class CrossAttnWeightsC(C.Structure):
    _fields_ = [(n, C.c_void_p) for n in ("norm_g", "norm_b", "text_norm_g", "text_norm_b", "wq", "bq", "wk", "bk", "wv", "bv",
                                           "sty_norm_g", "sty_norm_b", "sty_emb_w", "sty_emb_b", "sty_out_w", "sty_out_b")]


class SamplerOptsC(C.Structure):
    _fields_ = [("kind", C.c_int32), ("diffusion_steps", C.c_int32), ("respacing", C.c_int32),
                ("jump_length", C.c_int32), ("jump_n_sample", C.c_int32), ("overlap_len", C.c_int32),
                ("add_blend", C.c_int32), ("no_resample", C.c_int32), ("no_repaint", C.c_int32),
                ("clip_denoised", C.c_int32), ("noise_mode", C.c_int32), ("seed", C.c_uint64),
                ("same_overlap_noisy", C.c_int32), ("clip_idx", C.c_int32), ("eta", C.c_float)]


# every symbol include/diffsheg_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
SYMBOLS = {
    "dsh_last_error": (C.c_char_p, []),
    "dsh_version": (C.c_char_p, []),
    "dsh_create": (C.c_int, [C.POINTER(ModelConfigC), _P, C.POINTER(_P)]),
    "dsh_destroy": (C.c_int, [_P]),
    "dsh_load_tensor": (C.c_int, [_P, C.c_char_p, _P, C.POINTER(C.c_int64), C.c_int32]),
    "dsh_finalize_weights": (C.c_int, [_P]),
    "dsh_weight_bytes": (C.c_int64, [_P]),
    "dsh_set_condition": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P, _P]),
    "dsh_eval": (C.c_int, [_P, _P, _P, _P, _P, _P]),
    "dsh_eval_flops": (C.c_double, [_P]),
    "dsh_profile_enable": (C.c_int, [_P, C.c_int32]),
    "dsh_profile_read": (C.c_int, [_P, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "dsh_profile_class_info": (C.c_int, [_P, C.c_int32, C.POINTER(C.c_char_p), C.POINTER(C.c_char_p)]),
    "dsh_debug_copy": (C.c_int, [_P, C.c_char_p, _P]),
    "dsh_sample_num_draws": (C.c_int64, [C.POINTER(SamplerOptsC), C.c_int32, C.c_int32]),
    "dsh_sample_num_steps": (C.c_int64, [C.POINTER(SamplerOptsC), C.c_int32]),
    "dsh_sample": (C.c_int, [_P, C.POINTER(SamplerOptsC), _P, C.c_int32, _P, _P, C.c_int32, _P, C.c_int64, _P]),
    "dsh_sample_set_row_keys": (C.c_int, [_P, C.POINTER(C.c_uint64), C.c_int32]),
    "dsh_diffusion_table": (C.c_int32, [C.c_int32, C.c_int32, C.c_char_p, C.POINTER(C.c_double), C.c_int32]),
    "dsh_timestep_map": (C.c_int32, [C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32]),
    "dsh_jump_schedule": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32), C.c_int32]),
    "dsh_interp_time": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32]),
    "dsh_inv_standardize": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P, _P, _P]),
    "dsh_op_gemm": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32]),
    "dsh_op_gemm_f32_pro": (C.c_int, [_P, C.c_int32] + [_P, C.c_int32, C.c_int32] * 4 + [C.c_int32, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, C.c_int32, _P]),
    "dsh_debug_last_tl_variant": (C.c_int32, []),
    "dsh_op_tl_linear": (C.c_int, [_P, C.c_int32, _P, _P, _P, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32, _P, _P, _P, C.c_int32, C.c_int32, C.c_int32]),
    "dsh_op_tl2_ffn": (C.c_int, [_P] * 12 + [C.c_int32, C.c_int32, _P, C.c_int32, _P, _P, C.c_int32]),
    "dsh_op_cross_attention": (C.c_int, [_P, C.POINTER(CrossAttnWeightsC), _P, _P, _P] + [C.c_int32] * 7 + [_P]),
    "dsh_op_linear_attention": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "dsh_op_linear_attention_bf16": (C.c_int, [_P, _P, C.c_int32, C.c_int32, C.c_int32, C.c_int32, _P]),
    "dsh_op_layernorm": (C.c_int, [_P, _P, C.c_int32, C.c_int32, _P, _P, _P]),
    "dsh_op_philox_randn": (C.c_int, [_P, _P, C.c_int64, C.c_uint64, C.c_uint64]),
    "dsh_op_philox_randn_rows": (C.c_int, [_P, _P, C.c_int32, C.c_int64, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64)]),
}

_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise DshError(f"{LIB_PATH} not found: the HIP extension is not built and there is no CPU fallback "
                           f"(run `make -C diffsheg_amd/csrc` or __graft_entry__.build())")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)          # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int, what: str = "") -> int:
    if rc < 0:
        msg = lib().dsh_last_error()
        raise DshError(f"{what or 'libdiffsheg_hip'} failed ({rc}): {msg.decode() if msg else '?'}")
    return rc
