"""ctypes binding of libpuresound_hip.so (the C ABI declared in include/puresound_hip.h).

There is no CPU fallback: if the library is missing or a call fails, a RuntimeError is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
# PURESOUND_HIP_LIB: an experimental build of the same library (tools/build_variant.sh); kernel experiments only
LIB_PATH = os.environ.get("PURESOUND_HIP_LIB") or os.path.join(_HERE, "libpuresound_hip.so")
ABI_VERSION = 21

PS_NORM_NONE, PS_NORM_GLOBAL, PS_NORM_AFFINE = 0, 1, 2
PS_ACT = {"linear": 0, "relu": 1, "sigmoid": 2}
PS_OUT = {"linear": 0, "sigmoid": 1, "none": 2}

_f = C.POINTER(C.c_float)
_d = C.POINTER(C.c_double)
_vp = C.c_void_p


class Prologue(C.Structure):
    _fields_ = [("norm", C.c_int), ("prelu", C.c_int), ("stats", _vp), ("parts", C.c_int),
                ("count", C.c_double), ("eps", C.c_float), ("gamma", _vp), ("beta", _vp), ("slope", _vp),
                ("pre_relu", C.c_int), ("post_tanh", C.c_int)]


class F16x2Range(C.Structure):
    _fields_ = [("w_exp", C.c_int), ("x_bound", C.c_float), ("x_amax", _vp), ("x_amax_parts", C.c_int), ("y_amax", _vp),
                ("amax_mul", C.c_float), ("amax_add", C.c_float)]


PS_MAX_CELLS = 8


class FilmCell(C.Structure):
    _fields_ = [(k, _vp) for k in ("x", "wt_pairs", "res_pairs", "y")]


class GatesCell(C.Structure):
    _fields_ = [(k, _vp) for k in ("xh", "wt_units", "bias_units", "c", "h")]


class ProjLnCell(C.Structure):
    _fields_ = [(k, _vp) for k in ("x", "wt", "bias", "gamma", "beta", "res", "y", "gamma2", "beta2", "y2", "x_copy")] + [
        ("eps", C.c_float), ("eps2", C.c_float)]


class LstmArgs(C.Structure):
    _fields_ = [("gx", _vp), ("whh_t", _vp), ("h0", _vp), ("c0", _vp), ("hout", _vp), ("h_last", _vp),
                ("c_last", _vp)] + [(k, C.c_int) for k in ("N", "H", "D", "Q", "q_stride", "steps", "step_stride",
                                                           "ldt", "ldq", "state_shift")]


class TcnBlock(C.Structure):
    _fields_ = [("C", C.c_int), ("H", C.c_int), ("P", C.c_int), ("dilation", C.c_int), ("causal", C.c_int),
                ("in_norm", C.c_int), ("dw_norm", C.c_int), ("pw_norm", C.c_int),
                ("in_wt", _vp), ("in_embed_w", _vp), ("E", C.c_int),
                ("in_gamma", _vp), ("in_beta", _vp), ("in_slope", _vp),
                ("dw_w", _vp), ("dw_b", _vp), ("dw_gamma", _vp), ("dw_beta", _vp), ("dw_slope", _vp),
                ("pw_wt", _vp), ("pw_b", _vp), ("pw_gamma", _vp), ("pw_beta", _vp), ("pw_slope", _vp),
                ("out_wt", _vp), ("out_b", _vp),
                ("gemm_planes", C.c_int), ("in_wb", _vp), ("pw_wb", _vp), ("out_wb", _vp), ("hidden_bf16", C.c_int),
                ("w_exp", C.c_int * 3), ("dw_gmax", C.c_float), ("dw_bmax", C.c_float), ("pw_gmax", C.c_float),
                ("pw_bmax", C.c_float), ("in_wf", _vp), ("pw_wf", _vp), ("out_wf", _vp)]


# name -> (restype, argtypes); every symbol include/puresound_hip.h declares
SIGNATURES = {
    "ps_abi_version": (C.c_int, []),
    "ps_last_error": (C.c_char_p, []),
    "ps_debug_flags": (C.c_int, [C.c_int]),
    "ps_debug_buffer": (C.c_int, [_vp]),
    "ps_profile_enable": (C.c_int, [C.c_int]),
    "ps_profile_read": (C.c_int, [C.c_char_p, C.POINTER(C.c_double), C.POINTER(C.c_int)]),
    "ps_stats_parts": (C.c_int, [C.c_int, C.c_int]),
    "ps_conv1x1_stats_parts": (C.c_int, [C.c_int, C.c_int]),
    "ps_dwconv_stats_parts": (C.c_int, [C.c_int, C.c_int]),
    "ps_padded_frames": (C.c_int, [C.c_int]),
    "ps_pad_rows_f32": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int, _vp]),
    "ps_unpad_rows_f32": (C.c_int, [_vp, _vp, C.c_int64, C.c_int, C.c_int, _vp]),
    "ps_free_encode_f32": (C.c_int, [_vp, _vp, _vp] + [C.c_int] * 8 + [_vp]),
    "ps_free_decode_f32": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp] + [C.c_int] * 7 + [_vp]),
    "ps_free_decode_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "ps_free_decode_ws_f32": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp] + [C.c_int] * 7 + [_vp, C.c_size_t, _vp]),
    "ps_free_decode_moments_parts": (C.c_int, [C.c_int] * 6),
    "ps_free_decode_moments_f32": (C.c_int, [_vp, _vp, C.c_int, _vp, _vp] + [C.c_int] * 7 + [_vp, C.c_int, C.c_int, _vp,
                                              _vp, C.c_size_t, _vp]),
    "ps_frame_f32": (C.c_int, [_vp, _vp] + [C.c_int] * 6 + [_vp]),
    "ps_complex_mask_f32": (C.c_int, [_vp, _vp, _vp] + [C.c_int] * 4 + [_vp]),
    "ps_polar_mask_f32": (C.c_int, [_vp, _vp, _vp] + [C.c_int] * 3 + [_vp]),
    "ps_magphase_f32": (C.c_int, [_vp, _vp] + [C.c_int] * 4 + [_vp]),
    "ps_istft_ola_f32": (C.c_int, [_vp, _vp, _vp] + [C.c_int] * 6 + [_vp]),
    "ps_conv1x1_f32": (C.c_int, [_vp, _vp, _vp] + [C.c_int] * 5 + [C.POINTER(Prologue), _vp, _vp, _vp, _vp, _vp]),
    "ps_conv1x1_bf16_weight_bytes": (C.c_size_t, [C.c_int] * 3),
    "ps_conv1x1_bf16_f32": (C.c_int, [_vp, _vp, _vp] + [C.c_int] * 6 + [C.POINTER(Prologue), _vp, _vp, _vp, _vp, _vp]),
    "ps_dwconv_f32": (C.c_int, [_vp, _vp, _vp, _vp] + [C.c_int] * 7 + [C.POINTER(Prologue), _vp, _vp]),
    "ps_dwconv_io": (C.c_int, [_vp, C.c_int, _vp, _vp, _vp] + [C.c_int] * 8 + [C.POINTER(Prologue), _vp, _vp]),
    "ps_dwconv_amax_ok": (C.c_int, [C.c_int] * 3),
    "ps_dwconv_amax_f32": (C.c_int, [_vp] * 4 + [C.c_int] * 7 + [C.POINTER(Prologue), _vp, _vp]),
    "ps_conv1x1_bf16_io": (C.c_int, [_vp, C.c_int, _vp, _vp] + [C.c_int] * 7 + [C.POINTER(Prologue), _vp, _vp, _vp, _vp, _vp]),
    "ps_conv1x1_f16x2_f32": (C.c_int, [_vp, _vp, C.POINTER(F16x2Range), _vp] + [C.c_int] * 5 + [C.POINTER(Prologue), _vp, _vp, _vp, _vp, _vp]),
    "ps_absmax_parts": (C.c_int, []),
    "ps_absmax_f32": (C.c_int, [_vp, _vp] + [C.c_int] * 4 + [_vp]),
    "ps_attn_stats_pool_f32": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _vp]),
    "ps_attn_stats_pool_len_f32": (C.c_int, [_vp, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, C.c_float, _vp]),
    "ps_row_stats_parts": (C.c_int, []),
    "ps_row_stats_f64": (C.c_int, [_vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "ps_attn_weights_f32": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "ps_lstm_f32": (C.c_int, [C.POINTER(LstmArgs), _vp]),
    "ps_rnn_f32": (C.c_int, [C.POINTER(LstmArgs), C.c_int, _vp, _vp]),
    "ps_lstm_f16x2_f32": (C.c_int, [C.POINTER(LstmArgs), _vp]),
    "ps_lstm_fmajor_ok": (C.c_int, [C.POINTER(LstmArgs), C.c_int]),
    "ps_lstm_fmajor_f16x2_f32": (C.c_int, [C.POINTER(LstmArgs), C.c_int, _vp]),
    "ps_lstm_fmajor_h256_ok": (C.c_int, [C.POINTER(LstmArgs), C.c_int]),
    "ps_lstm_fmajor_h256_f16x2_f32": (C.c_int, [C.POINTER(LstmArgs), C.c_int, _vp, C.POINTER(C.c_float), _vp]),
    "ps_lstm_fmajor_coop_workspace_bytes": (C.c_size_t, [C.POINTER(LstmArgs), C.c_int]),
    "ps_lstm_fmajor_coop_f16x2_f32": (C.c_int, [C.POINTER(LstmArgs), C.c_int, _vp, C.POINTER(C.c_float), _vp, C.c_size_t, _vp]),
    "ps_unfold_taps_f32": (C.c_int, [_vp, _vp] + [C.c_int] * 7 + [_vp, _vp, _vp, C.c_int, _vp]),
    "ps_unfold_taps_out_f32": (C.c_int, [_vp, _vp] + [C.c_int] * 8 + [_vp, _vp, _vp, C.c_int, _vp]),
    "ps_gated_product_f32": (C.c_int, [_vp, _vp, _vp] + [C.c_int] * 4 + [C.POINTER(Prologue), C.POINTER(Prologue), _vp]),
    "ps_segment_overlap_f32": (C.c_int, [_vp, _vp, C.c_int64] + [C.c_int] * 6 + [_vp]),
    "ps_film_conv_f32": (C.c_int, [_vp, _vp, _vp, _vp] + [C.c_int] * 4 + [_vp]),
    "ps_lstm_gates_cell_f32": (C.c_int, [_vp, _vp, _vp, _vp, _vp] + [C.c_int] * 6 + [_vp]),
    "ps_proj_layernorm_f32": (C.c_int, [_vp] * 5 + [C.c_float, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp] + [C.c_int] * 6
                              + [_vp]),
    "ps_film_conv_cells_f32": (C.c_int, [C.POINTER(FilmCell)] + [C.c_int] * 4 + [_vp]),
    "ps_lstm_gates_cell_cells_f32": (C.c_int, [C.POINTER(GatesCell)] + [C.c_int] * 6 + [_vp]),
    "ps_proj_layernorm_cells_f32": (C.c_int, [C.POINTER(ProjLnCell)] + [C.c_int] * 6 + [_vp]),
    "ps_proj_layernorm_amax_parts": (C.c_int, [C.c_int]),
    "ps_proj_layernorm_amax_f32": (C.c_int, [_vp] * 5 + [C.c_float, _vp, _vp, _vp, _vp, C.c_float, _vp, _vp] + [C.c_int] * 6
                                   + [_vp, _vp]),
    "ps_self_attention_f32": (C.c_int, [_vp, _vp] + [C.c_int] * 9 + [_vp]),
    "ps_add_position_f32": (C.c_int, [_vp, _vp, _vp] + [C.c_int] * 7 + [_vp]),
    "ps_overlap_average_f32": (C.c_int, [_vp, C.c_int, _vp, _vp, C.c_int, C.c_int, C.c_int, _vp]),
    "ps_stream_windows_f32": (C.c_int, [_vp, _vp, _vp] + [C.c_int] * 4 + [_vp]),
    "ps_stream_overlap_f32": (C.c_int, [_vp] * 5 + [C.c_int] * 4 + [_vp]),
    "ps_unfold2d_f32": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp] + [C.c_int] * 14 + [_vp]),
    "ps_conv2d_f32": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp] + [C.c_int] * 16 + [_vp, _vp]),
    "ps_conv2d_stats_parts": (C.c_int, [C.c_int] * 3),
    "ps_conv2d_f16x2_f32": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, C.c_int, _vp, _vp] + [C.c_int] * 16 + [_vp, _vp, _vp]),
    "ps_conv2d_stats_f32": (C.c_int, [_vp, C.c_int, _vp, C.c_int, _vp, _vp, _vp] + [C.c_int] * 15 + [_vp, _vp]),
    "ps_activation_f32": (C.c_int, [_vp, C.c_int, _vp, C.c_int64, C.c_int, C.c_int, _vp]),
    "ps_magnitude_f32": (C.c_int, [_vp, _vp] + [C.c_int] * 6 + [_vp]),
    "ps_fill_span_f32": (C.c_int, [_vp, _vp] + [C.c_int] * 6 + [C.c_float, _vp]),
    "ps_real_mask_f32": (C.c_int, [_vp, _vp, _vp, C.c_int64, C.c_int, C.c_int, _vp]),
    "ps_norm_activation_f32": (C.c_int, [_vp, C.POINTER(Prologue), C.c_double, C.c_double, C.c_int, C.c_int, _vp]
                               + [C.c_int] * 4 + [_vp]),
    "ps_add_f32": (C.c_int, [_vp, _vp, _vp, C.c_int64, _vp]),
    "ps_lstm_cell_f32": (C.c_int, [_vp, _vp, _vp] + [C.c_int] * 6 + [_vp]),
    "ps_chan_layernorm_f32": (C.c_int, [_vp, _vp, _vp, C.c_float, _vp, C.c_int, _vp, _vp, _vp] + [C.c_int] * 4 + [_vp]),
    "ps_film_apply_f32": (C.c_int, [_vp, _vp, _vp] + [C.c_int] * 4 + [_vp]),
    "ps_embed_bias_f32": (C.c_int, [_vp, _vp, _vp, C.c_int, C.c_int, C.c_int, C.c_int, _vp]),
    "ps_wave_moments_chunks": (C.c_int, [C.c_int]),
    "ps_wave_moments_f64": (C.c_int, [_vp, _vp, _vp] + [C.c_int] * 4 + [_vp]),
    "ps_conv_tasnet_workspace_bytes": (C.c_size_t, [C.c_int] * 4),
    "ps_conv_tasnet_f32": (C.c_int, [C.POINTER(TcnBlock), C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                     C.c_int, _vp, C.c_size_t, _vp]),
    "ps_conv_tasnet_ranged_f32": (C.c_int, [C.POINTER(TcnBlock), C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                            C.c_int, _vp, C.c_size_t, _vp, C.c_int, _vp]),
    "ps_conv1x1_f16_rows_ok": (C.c_int, [C.c_int] * 4),
    "ps_conv1x1_f16x2_ln_ok": (C.c_int, [C.c_int] * 4),
    "ps_conv1x1_f16x2_ln_f32": (C.c_int, [_vp, _vp, C.POINTER(F16x2Range), _vp] + [C.c_int] * 5
                                + [C.POINTER(Prologue), _vp, _vp, _vp, C.c_float, _vp, C.c_int, _vp]),
    "ps_conv1x1_f16x2_fmajor_ok": (C.c_int, [C.c_int] * 6),
    "ps_conv1x1_f16x2_fmajor_f32": (C.c_int, [_vp, _vp, C.POINTER(F16x2Range), _vp] + [C.c_int] * 6 + [_vp, _vp]),
    "ps_conv1x1_f16_rows": (C.c_int, [_vp, _vp, C.POINTER(F16x2Range), _vp] + [C.c_int] * 5 + [C.POINTER(Prologue), _vp, _vp, _vp, _vp, _vp]),
    "ps_conv_tasnet_bf16_rows": (C.c_int, [C.POINTER(TcnBlock), C.c_int, _vp, _vp, _vp, C.c_int, C.c_int, C.c_int,
                                           C.c_int, _vp, C.c_size_t, _vp]),
}

_lib: Optional[C.CDLL] = None


def lib() -> C.CDLL:
    """Load (once) and return the HIP library; raise if it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                f"or `make -C puresound_amd/csrc`. puresound_amd has no CPU fallback.")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(handle, name)  # AttributeError if a declared symbol is not exported
            fn.restype = res
            fn.argtypes = args
        if handle.ps_abi_version() != ABI_VERSION:
            raise RuntimeError(f"libpuresound_hip.so ABI {handle.ps_abi_version()} != binding {ABI_VERSION}")
        _lib = handle
    return _lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = lib().ps_last_error().decode(errors="replace")
        raise RuntimeError(f"{what} failed (rc={rc}): {msg}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


def stream_ptr(device: torch.device) -> int:
    return torch.cuda.current_stream(device).cuda_stream


def require_device(t: torch.Tensor, what: str, allow_bf16: bool = False) -> None:
    """The product path is HIP only: refuse CPU tensors loudly instead of falling back."""
    if not t.is_cuda:
        raise RuntimeError(f"{what}: puresound_amd runs on a ROCm device only (got a {t.device} tensor); "
                           f"there is no CPU fallback")
    if t.dtype != torch.float32 and not (allow_bf16 and t.dtype == torch.bfloat16):
        raise RuntimeError(f"{what}: fp32 tensors only (got {t.dtype})")


def require_weight(w: torch.Tensor, like: torch.Tensor, what: str) -> None:
    """A parameter handed to a kernel as a raw pointer: it has to live on the input's device, as contiguous fp32 --
    a model left on the CPU, on another GPU or in half precision would otherwise reach the kernel as a wild pointer."""
    if not w.is_cuda or w.device != like.device:
        raise RuntimeError(f"{what}: the module's parameters are on {w.device} but the input is on {like.device} "
                           f"(move the model with .to(device); there is no CPU fallback)")
    if w.dtype != torch.float32:
        raise RuntimeError(f"{what}: fp32 parameters only (got {w.dtype})")
    if not w.is_contiguous():
        raise RuntimeError(f"{what}: parameters must be contiguous")


def padded_frames(t: int) -> int:
    """Same law as ps_padded_frames: an odd multiple of 128 frames >= t (no power-of-two row stride)."""
    tiles = (t + 127) // 128
    if tiles % 2 == 0:
        tiles += 1
    return tiles * 128
