"""ctypes binding of libadaface_hip.so (include/adaface_hip.h).

The library is the product: if it is missing or fails to load this module raises —
there is no CPU or eager-PyTorch fallback anywhere in adaface_amd.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

AF_DTYPE_BF16 = 0
AF_DTYPE_F32 = 1
DTYPES = {"bf16": AF_DTYPE_BF16, "bfloat16": AF_DTYPE_BF16, "f32": AF_DTYPE_F32, "fp32": AF_DTYPE_F32,
          "float32": AF_DTYPE_F32}

_LIB_PATH = Path(__file__).resolve().parent / "libadaface_hip.so"


class AfConfig(C.Structure):
    """struct af_config (include/adaface_hip.h)."""
    _fields_ = [
        ("dtype", C.c_int),
        ("build_unet", C.c_int),
        ("in_channels", C.c_int), ("model_channels", C.c_int), ("out_channels", C.c_int),
        ("num_res_blocks", C.c_int),
        ("n_attention_resolutions", C.c_int), ("attention_resolutions", C.c_int * 8),
        ("n_channel_mult", C.c_int), ("channel_mult", C.c_int * 8),
        ("num_heads", C.c_int), ("context_dim", C.c_int), ("transformer_depth", C.c_int),
        ("n_context_layers", C.c_int),
        ("build_vae", C.c_int),
        ("vae_ch", C.c_int), ("vae_out_ch", C.c_int), ("vae_num_res_blocks", C.c_int),
        ("vae_z_channels", C.c_int), ("vae_embed_dim", C.c_int),
        ("n_vae_ch_mult", C.c_int), ("vae_ch_mult", C.c_int * 8),
        ("build_vae_encoder", C.c_int), ("vae_in_channels", C.c_int),
        ("build_clip", C.c_int),
        ("clip_vocab", C.c_int), ("clip_hidden", C.c_int), ("clip_layers", C.c_int), ("clip_heads", C.c_int),
        ("clip_intermediate", C.c_int), ("clip_max_pos", C.c_int),
    ]


class AfError(RuntimeError):
    pass


_lib = None

# every symbol include/adaface_hip.h declares: (name, restype, argtypes)
_P = C.c_void_p
_SIGS = [
    ("af_last_error", C.c_char_p, []),
    ("af_version", C.c_int, []),
    ("af_create", C.c_int, [C.c_int, C.POINTER(AfConfig), C.POINTER(_P)]),
    ("af_destroy", None, [_P]),
    ("af_load_tensor", C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64)]),
    ("af_load_tensor_device", C.c_int, [_P, C.c_char_p, _P, C.c_int, C.POINTER(C.c_int64)]),
    ("af_num_tensors", C.c_int, [_P]),
    ("af_tensor_name", C.c_char_p, [_P, C.c_int]),
    ("af_tensor_loaded", C.c_int, [_P, C.c_int]),
    ("af_tensor_shape", C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    ("af_set_context", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    ("af_unet_forward", C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    ("af_unet_forward_twin", C.c_int, [_P, _P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    ("af_ddim_step", C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                               C.c_float, _P, _P, _P]),
    ("af_lincomb", C.c_int, [_P, C.c_int64, _P, C.c_float, _P, C.c_float, _P, C.c_float, _P, C.c_float, C.c_int, _P]),
    ("af_vae_decode", C.c_int, [_P, _P, C.c_float, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    ("af_to_uint8", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int, _P]),
    ("af_arena_bytes", C.c_int64, [_P]),
    ("af_prof_enable", C.c_int, [C.c_int]),
    ("af_prof_reset", C.c_int, []),
    ("af_set_conv_attn", C.c_int, [_P, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("af_vae_encode", C.c_int, [_P, _P, _P, C.c_int, C.c_int, C.c_int, _P]),
    ("af_posterior_sample", C.c_int, [_P, _P, C.c_float, _P, C.c_int, C.c_int, C.c_int, _P]),
    ("af_prof_set_stride", C.c_int, [C.c_int]),
    ("af_last_gemm_plan", C.c_int, [C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("af_prof_collect", C.c_int, [C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64), C.POINTER(C.c_double),
                                  C.POINTER(C.c_double)]),
    ("af_prof_event_overhead_us", C.c_double, [_P, C.c_int]),
    ("af_unet_num_blocks", C.c_int, [_P]),
    ("af_unet_block_shape", C.c_int, [_P, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    ("af_unet_set_tap", C.c_int, [_P, C.c_int, _P]),
    ("af_gemm_plan_counts", C.c_int, [C.POINTER(C.c_int64)]),
    ("af_gemm_plan_counts_reset", C.c_int, []),
    ("af_knob_set", C.c_int, [C.c_char_p, C.c_int]),
    ("af_knob_get", C.c_int, [C.c_char_p, C.POINTER(C.c_int)]),
    ("af_knob_reset", C.c_int, []),
    ("af_op_conv2d", C.c_int, [C.c_int, _P, _P, _P, _P, _P] + [C.c_int] * 9 + [_P]),
    ("af_op_linear", C.c_int, [C.c_int, _P, _P, _P, _P, _P, C.c_int64, C.c_int, C.c_int, C.c_int, _P]),
    ("af_op_groupnorm", C.c_int, [C.c_int, _P, _P, _P, C.c_float, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    ("af_op_conv_gn", C.c_int, [_P, _P, _P, _P, _P, _P, C.c_float, C.c_int, _P, _P] + [C.c_int] * 5 + [_P]),
    ("af_set_fp8", C.c_int, [_P, C.c_int]),
    ("af_fp8_gemm_launches", C.c_int64, []),
    ("af_halo8_launches", C.c_int64, []),
    ("af_rowpanel_launches", C.c_int64, []),
    ("af_up_phase4_launches", C.c_int64, []),
    ("af_gn_producer_launches", C.c_int64, []),
    ("af_attn_short_launches", C.c_int64, []),
    ("af_gn_consumer_launches", C.c_int64, []),
    ("af_xattn_fused_launches", C.c_int64, []),
    ("af_op_xattn_fused", C.c_int, [C.c_void_p] * 10 + [C.c_int] * 3 + [C.c_void_p]),
    ("af_op_conv2d_fp8", C.c_int, [_P, _P, _P, _P, _P] + [C.c_int] * 10 + [_P]),
    ("af_op_groupnorm_fp8", C.c_int, [_P, _P, _P, C.c_float, C.c_int, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, _P]),
    ("af_op_layernorm_fp8", C.c_int, [_P, _P, _P, C.c_float, _P, C.c_int64, C.c_int, C.c_int, _P]),
    ("af_op_layernorm", C.c_int, [C.c_int, _P, _P, _P, C.c_float, _P, C.c_int64, C.c_int, _P]),
    ("af_op_attention", C.c_int, [C.c_int, _P, _P, _P, _P, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, C.c_int, _P]),
    ("af_clip_embed_tokens", C.c_int, [_P, _P, C.c_int64, _P, _P]),
    ("af_clip_text_forward", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, C.c_float, _P, _P]),
    ("af_clip_text_forward3", C.c_int, [_P, _P, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, _P, _P]),
    ("af_op_timestep_embedding", C.c_int, [C.c_int, _P, _P, C.c_int, C.c_int, _P]),
    ("af_op_gn_conv1x1", C.c_int, [_P, _P, _P, C.c_float, _P, _P, _P, _P] + [C.c_int] * 5 + [_P]),
    ("af_flops_issued", C.c_double, [C.c_int]),
    ("af_clock_probe", C.c_int, [_P, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
]
EXPORTED_SYMBOLS = [s[0] for s in _SIGS]


def lib_path() -> Path:
    return _LIB_PATH


def load():
    """Load the shared library (once) and declare every entry point's signature."""
    global _lib
    if _lib is not None:
        return _lib
    if not _LIB_PATH.exists():
        raise AfError(
            f"{_LIB_PATH} not found: build it with `python -m adaface_amd.build` (needs hipcc). "
            "adaface_amd has no fallback path.")
    lib = C.CDLL(os.fspath(_LIB_PATH))
    for name, res, args in _SIGS:
        fn = getattr(lib, name)  # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def set_knob(name: str, value: int) -> None:
    """af_knob_set: force a planner choice / kernel variant (tests, lab scripts)."""
    check(load().af_knob_set(name.encode(), int(value)), f"af_knob_set({name})")


def reset_knobs() -> None:
    load().af_knob_reset()


def plan_counts(reset: bool = False) -> dict:
    """af_gemm_plan_counts as a dict: tile0..tile5, halo, splitk, ln_consumer, ln_producer."""
    lib = load()
    c = (C.c_int64 * 10)()
    check(lib.af_gemm_plan_counts(c), "af_gemm_plan_counts")
    out = {f"tile{i}": int(c[i]) for i in range(6)}
    out["halo"], out["splitk"], out["ln_consumer"], out["ln_producer"] = int(c[6]), int(c[7]), int(c[8]), int(c[9])
    out["fp8"] = int(lib.af_fp8_gemm_launches())
    out["halo8"] = int(lib.af_halo8_launches())
    out["rowpanel"] = int(lib.af_rowpanel_launches())
    out["up_phase4"] = int(lib.af_up_phase4_launches())
    out["gn_producer"] = int(lib.af_gn_producer_launches())
    out["attn_short"] = int(lib.af_attn_short_launches())
    out["gn_consumer"] = int(lib.af_gn_consumer_launches())
    out["xattn_fused"] = int(lib.af_xattn_fused_launches())
    if reset:
        lib.af_gemm_plan_counts_reset()
    return out


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = load().af_last_error().decode("utf-8", "replace")
        raise AfError(f"{what or 'libadaface_hip'} failed (rc={rc}): {msg}")


def stream_ptr():
    """Current torch HIP stream as a void* for the C ABI."""
    import torch
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """Device/host data pointer of a tensor (None -> NULL)."""
    return C.c_void_p(0 if t is None else t.data_ptr())
