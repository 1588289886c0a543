"""Tensor-level front of the C ABI: each function enqueues exactly one entry point of
libpuresound_hip.so on the current HIP stream of the tensors' device.  PyTorch is used for device
memory and streams only; all arithmetic happens in the HIP kernels.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional, Sequence

import torch

from . import _abi
from ._abi import (F16x2Range, LstmArgs, Prologue, TcnBlock, check, lib, padded_frames, ptr, require_device, require_weight,
                   stream_ptr)


def pack_wt(w: torch.Tensor) -> torch.Tensor:
    """[M,K] (or [M,K,1]) conv weight -> kernel layout [ceil(M/256)][ceil16(K)][256]: transposed (k-major),
    zero padded, one 256-channel output tile after the other (see ps_conv1x1_f32)."""
    if w.dim() == 3:
        w = w[:, :, 0]
    m, k = w.shape
    kp, mt = (k + 15) // 16 * 16, (m + 255) // 256
    out = torch.zeros(mt, kp, 256, dtype=torch.float32, device=w.device)
    wt = w.detach().to(torch.float32).t()  # [K, M]
    for i in range(mt):
        cols = min(256, m - i * 256)
        out[i, :k, :cols] = wt[:, i * 256:i * 256 + cols]
    return out


def pad_rows(x: torch.Tensor, min_frames: int = 0) -> torch.Tensor:
    """compact [..., T] -> padded [..., ldt] (zeros in the pad); ldt also covers `min_frames`."""
    require_device(x, "pad_rows")
    x = x.contiguous()
    t = x.shape[-1]
    ldt = padded_frames(max(t, min_frames))
    out = torch.empty(*x.shape[:-1], ldt, dtype=torch.float32, device=x.device)
    rows = x.numel() // t
    check(lib().ps_pad_rows_f32(ptr(x), ptr(out), rows, t, ldt, stream_ptr(x.device)), "ps_pad_rows_f32")
    return out


def unpad_rows(x: torch.Tensor, t: int) -> torch.Tensor:
    require_device(x, "unpad_rows")
    ldt = x.shape[-1]
    out = torch.empty(*x.shape[:-1], t, dtype=torch.float32, device=x.device)
    rows = x.numel() // ldt
    check(lib().ps_unpad_rows_f32(ptr(x), ptr(out), rows, t, ldt, stream_ptr(x.device)), "ps_unpad_rows_f32")
    return out


def free_encode(wav: torch.Tensor, w: torch.Tensor, hop: int, relu: bool = False,
                min_frames=None) -> tuple[torch.Tensor, int]:
    """wav [N,L], w [C,1,win] -> (feats padded [N,C,ldt], T).  `min_frames`: int or callable T -> frames the
    consumer needs the (zero-filled) rows to hold (segment padding of the dual-path maskers)."""
    require_device(wav, "free_encode")
    require_weight(w, wav, "free_encode")
    wav = wav.contiguous()
    n, length = wav.shape
    c, _, win = w.shape
    if length < win:
        raise RuntimeError(f"free_encode: input length {length} is shorter than the window {win}")
    t = (length - win) // hop + 1
    need = min_frames(t) if callable(min_frames) else (min_frames or 0)
    ldt = padded_frames(max(t, need))
    # frames beyond T: zero when a consumer asked for them (segment padding: only those columns are cleared, the kernel
    # writes [0, T) and nothing else), otherwise never read as data
    feats = torch.empty(n, c, ldt, dtype=torch.float32, device=wav.device)
    if need > t:
        feats[..., t:].zero_()
    check(lib().ps_free_encode_f32(ptr(wav), ptr(w), ptr(feats), n, length, c, win, hop, t, ldt, int(relu),
                                   stream_ptr(wav.device)), "ps_free_encode_f32")
    return feats, t


def free_decode(feats: torch.Tensor, t: int, w: torch.Tensor, hop: int, mask: Optional[torch.Tensor] = None,
                mask_act: str = "linear", out_mode: str = "none", out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """feats/mask padded [N,C,ldt] -> waveform [N,(T-1)*hop+win] (into `out` if given: contiguous rows)."""
    require_device(feats, "free_decode")
    require_weight(w, feats, "free_decode")
    n, c, ldt = feats.shape
    win = w.shape[-1]
    if out is None:
        out = torch.empty(n, (t - 1) * hop + win, dtype=torch.float32, device=feats.device)
    elif tuple(out.shape) != (n, (t - 1) * hop + win) or not out.is_contiguous():
        raise RuntimeError("free_decode: `out` must be a contiguous [N, (T-1)*hop+win] tensor")
    # the matrix-pipe decoder (win = 32, hop = 16, long rows) completes tile boundaries through a small side buffer
    ws_bytes = lib().ps_free_decode_workspace_bytes(n, t, win, hop)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=feats.device) if ws_bytes else None
    check(lib().ps_free_decode_ws_f32(ptr(feats), ptr(mask), _abi.PS_ACT[mask_act], ptr(w), ptr(out), n, c, t, ldt,
                                      win, hop, _abi.PS_OUT[out_mode], ptr(ws), ws_bytes, stream_ptr(feats.device)),
          "ps_free_decode_ws_f32")
    return out


def align_reference(ref: torch.Tensor, length: int) -> torch.Tensor:
    """_align_waveform's rule for the reference (base_nn.py:398-412): shorter than the estimate it is left-padded with
    zeros, longer it is cut."""
    have = ref.shape[-1]
    if have < length:
        return torch.nn.functional.pad(ref, (length - have, 0))
    return ref[..., :length]


def free_decode_moments(feats: torch.Tensor, t: int, w: torch.Tensor, hop: int, ref: torch.Tensor,
                        mask: Optional[torch.Tensor] = None, mask_act: str = "linear", out_mode: str = "none",
                        out: Optional[torch.Tensor] = None) -> tuple[torch.Tensor, torch.Tensor]:
    """free_decode + the moments [N, 5] (fp64: sum a, sum b, sum a^2, sum b^2, sum ab over the (T-1)*hop+win output
    samples) of the estimate a and the reference b = ref [N, L_ref] aligned by the reference's rule (align_reference).
    One pass: the moments are the decoder's epilogue (ps_free_decode_moments_f32) where that kernel exists, otherwise
    the decoder followed by ps_wave_moments_f64."""
    require_device(feats, "free_decode_moments")
    require_device(ref, "free_decode_moments")
    require_weight(w, feats, "free_decode_moments")
    n, c, ldt = feats.shape
    win = w.shape[-1]
    lout = (t - 1) * hop + win
    if ref.dim() != 2 or ref.shape[0] != n or ref.dtype != torch.float32 or ref.shape[1] < 1:
        raise RuntimeError("free_decode_moments: `ref` must be an fp32 [N, L_ref] tensor")
    parts = lib().ps_free_decode_moments_parts(n, c, t, ldt, win, hop)
    if parts == 0:
        out = free_decode(feats, t, w, hop, mask, mask_act, out_mode, out)
        return out, wave_moments(out, align_reference(ref, lout))
    if out is None:
        out = torch.empty(n, lout, dtype=torch.float32, device=feats.device)
    elif tuple(out.shape) != (n, lout) or not out.is_contiguous():
        raise RuntimeError("free_decode_moments: `out` must be a contiguous [N, (T-1)*hop+win] tensor")
    if ref.stride(1) != 1:
        ref = ref.contiguous()
    ws_bytes = lib().ps_free_decode_workspace_bytes(n, t, win, hop)
    ws = torch.empty(ws_bytes // 4, dtype=torch.float32, device=feats.device)
    part = torch.empty(n, parts, 5, dtype=torch.float64, device=feats.device)
    check(lib().ps_free_decode_moments_f32(ptr(feats), ptr(mask), _abi.PS_ACT[mask_act], ptr(w), ptr(out), n, c, t, ldt,
                                           win, hop, _abi.PS_OUT[out_mode], ptr(ref), ref.stride(0), ref.shape[1],
                                           ptr(part), ptr(ws), ws_bytes, stream_ptr(feats.device)),
          "ps_free_decode_moments_f32")
    return out, part.sum(dim=1)


def frame(wav: torch.Tensor, win: int, hop: int) -> tuple[torch.Tensor, int]:
    """wav [N,L] -> (frames padded [N,win,ldt], T): frames[n][k][t] = wav[n][t*hop+k]."""
    require_device(wav, "frame")
    wav = wav.contiguous()
    n, length = wav.shape
    if length < win:
        raise RuntimeError(f"frame: input length {length} is shorter than the window {win}")
    t = (length - win) // hop + 1
    ldt = padded_frames(t)
    out = torch.empty(n, win, ldt, dtype=torch.float32, device=wav.device)
    check(lib().ps_frame_f32(ptr(wav), ptr(out), n, length, win, hop, t, ldt, stream_ptr(wav.device)), "ps_frame_f32")
    return out, t


def complex_mask(feats: torch.Tensor, mask: torch.Tensor, mask_act: str = "linear") -> torch.Tensor:
    """[re;im] channel halves, padded [N,2H,ldt] x mask [N,2H,ldt] -> complex product, same layout."""
    require_device(feats, "complex_mask")
    n, c2, ldt = feats.shape
    if c2 % 2 or mask.shape != feats.shape:
        raise RuntimeError("complex_mask: feats and mask must be [N, 2*half, ldt] with equal shapes")
    out = torch.empty_like(feats)
    check(lib().ps_complex_mask_f32(ptr(feats), ptr(mask), ptr(out), n, c2 // 2, ldt, _abi.PS_ACT[mask_act],
                                    stream_ptr(feats.device)), "ps_complex_mask_f32")
    return out


def polar_mask(feats: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """[re;im] channel halves, padded [N,2H,ldt], x mask in the same layout -> polar-form product (base_nn.py:161-190)."""
    require_device(feats, "polar_mask")
    n, c2, ldt = feats.shape
    if c2 % 2 or mask.shape != feats.shape:
        raise RuntimeError("polar_mask: feats and mask must be [N, 2*half, ldt] with equal shapes")
    out = torch.empty_like(feats)
    check(lib().ps_polar_mask_f32(ptr(feats), ptr(mask), ptr(out), n, c2 // 2, ldt, stream_ptr(feats.device)),
          "ps_polar_mask_f32")
    return out


def magphase(spec: torch.Tensor, take_sqrt: bool) -> torch.Tensor:
    """analysis product [re;im] padded [N,2H,ldt] -> [mags;phase] in the same layout (lobe/encoder.py:384-389)."""
    require_device(spec, "magphase")
    n, c2, ldt = spec.shape
    out = torch.empty_like(spec)
    check(lib().ps_magphase_f32(ptr(spec), ptr(out), n, c2 // 2, ldt, int(take_sqrt), stream_ptr(spec.device)),
          "ps_magphase_f32")
    return out


def istft_ola(frames: torch.Tensor, t: int, window: torch.Tensor, hop: int, out_mode: str = "none") -> torch.Tensor:
    """synthesis frames padded [N,n_fft,ldt] -> waveform [N,(T-1)*hop+n_fft] (window, /n_fft, OLA, /window-sum)."""
    require_device(frames, "istft_ola")
    n, n_fft, ldt = frames.shape
    out = torch.empty(n, (t - 1) * hop + n_fft, dtype=torch.float32, device=frames.device)
    check(lib().ps_istft_ola_f32(ptr(frames), ptr(window), ptr(out), n, n_fft, hop, t, ldt, _abi.PS_OUT[out_mode],
                                 stream_ptr(frames.device)), "ps_istft_ola_f32")
    return out


def make_prologue(norm: int = 0, prelu: bool = False, stats: Optional[torch.Tensor] = None, count: float = 0.0,
                  eps: float = 1e-8, gamma: Optional[torch.Tensor] = None, beta: Optional[torch.Tensor] = None,
                  slope: Optional[torch.Tensor] = None, pre_relu: bool = False, post_tanh: bool = False) -> Prologue:
    p = Prologue()
    p.pre_relu, p.post_tanh = int(pre_relu), int(post_tanh)
    p.norm, p.prelu = norm, int(prelu)
    p.stats = ptr(stats)
    p.parts = 0 if stats is None else stats.shape[1]
    p.count, p.eps = float(count), float(eps)
    p.gamma, p.beta, p.slope = ptr(gamma), ptr(beta), ptr(slope)
    return p


def conv1x1(x: torch.Tensor, t: int, wt: torch.Tensor, m: int, pro: Optional[Prologue] = None,
            bias: Optional[torch.Tensor] = None, bias_n: Optional[torch.Tensor] = None,
            res: Optional[torch.Tensor] = None, want_stats: bool = False,
            out: Optional[torch.Tensor] = None) -> tuple[torch.Tensor, Optional[torch.Tensor]]:
    """x padded [N,K,ldt], wt packed (pack_wt) -> y padded [N,M,ldt] (+ partial stats [N,parts,2] fp64)."""
    require_device(x, "conv1x1")
    n, k, ldt = x.shape
    y = out if out is not None else torch.zeros(n, m, ldt, dtype=torch.float32, device=x.device)
    stats = None
    if want_stats:
        parts = lib().ps_conv1x1_stats_parts(m, t)
        stats = torch.zeros(n, parts, 2, dtype=torch.float64, device=x.device)
    check(lib().ps_conv1x1_f32(ptr(x), ptr(wt), ptr(y), n, k, m, t, ldt, C.byref(pro) if pro is not None else None,
                               ptr(bias), ptr(bias_n), ptr(res), ptr(stats), stream_ptr(x.device)), "ps_conv1x1_f32")
    return y, stats


def pack_wt_bf16(w: torch.Tensor, planes: int) -> torch.Tensor:
    """[M,K] (or [M,K,1]) fp32 weight -> bf16 plane image [ceil(M/256)][ceil(K/16)][planes][256][16] for
    ps_conv1x1_bf16_f32: plane p holds bf16 of what the planes before it left over (planes = 1: plain rounding)."""
    if w.dim() == 3:
        w = w[:, :, 0]
    m, k = w.shape
    mt, ks = (m + 255) // 256, (k + 15) // 16
    rest = torch.zeros(mt * 256, ks * 16, dtype=torch.float32, device=w.device)
    rest[:m, :k] = w.detach().float()
    out = torch.empty(mt, ks, planes, 256, 16, dtype=torch.bfloat16, device=w.device)
    for p in range(planes):
        h = rest.to(torch.bfloat16)
        out[:, :, p] = h.reshape(mt, 256, ks, 16).permute(0, 2, 1, 3)
        rest = rest - h.float()
    return out.contiguous()


def pack_wt_f16x2(w: torch.Tensor) -> tuple[torch.Tensor, int]:
    """[M,K] (or [M,K,1]) fp32 weight -> (two-plane fp16 image [ceil(M/256)][ceil(K/16)][2][256][16] of 2^w_exp * W, w_exp)
    for ps_conv1x1_f16x2_f32: plane 0 = fp16(w'), plane 1 = fp16(w' - plane 0); w_exp puts max |w'| into [2^13, 2^14).
    (Reads the maximum back to the host: plan-build time only.)"""
    if w.dim() == 3:
        w = w[:, :, 0]
    m, k = w.shape
    wmax = float(w.detach().abs().max())
    if not (wmax < float("inf")):
        raise ValueError("pack_wt_f16x2: the weight holds inf / NaN")
    import math
    w_exp = 13 - math.frexp(wmax)[1] + 1 if wmax > 0 else 0  # frexp: wmax = f * 2^e, f in [0.5, 1)
    mt, ks = (m + 255) // 256, (k + 15) // 16
    rest = torch.zeros(mt * 256, ks * 16, dtype=torch.float32, device=w.device)
    rest[:m, :k] = torch.ldexp(w.detach().float(), torch.tensor(w_exp, device=w.device))
    out = torch.empty(mt, ks, 2, 256, 16, dtype=torch.float16, device=w.device)
    for p in range(2):
        h = rest.to(torch.float16)
        out[:, :, p] = h.reshape(mt, 256, ks, 16).permute(0, 2, 1, 3)
        rest = rest - h.float()
    return out.contiguous(), w_exp


def conv1x1_f16x2(x: torch.Tensor, t: int, wt_planes: torch.Tensor, w_exp: int, m: int,
                  pro: Optional[Prologue] = None, bias: Optional[torch.Tensor] = None,
                  bias_n: Optional[torch.Tensor] = None, res: Optional[torch.Tensor] = None,
                  want_stats: bool = False, out: Optional[torch.Tensor] = None, x_bound: float = 0.0,
                  x_amax: Optional[torch.Tensor] = None, want_amax: bool = False,
                  amax_map: Optional[tuple] = None):
    """ps_conv1x1_f32's contract in the fp16x2 arithmetic (ps_conv1x1_f16x2_f32; weights from pack_wt_f16x2).
    x_bound / x_amax: how the activations are brought into fp16's range (see include/puresound_hip.h); x_amax is an
    [N, parts] tensor of partial maxima (absmax(), or the y_amax of the producing launch); amax_map = (mul, add): the
    maxima are those of x in front of an affine prologue whose output is bounded by mul * max|x| + add.
    Returns (y, stats, y_amax)."""
    require_device(x, "conv1x1_f16x2")
    n, k, ldt = x.shape
    y = out if out is not None else torch.empty(n, m, ldt, dtype=torch.float32, device=x.device)
    stats = amax = None
    if want_stats:
        parts = lib().ps_conv1x1_stats_parts(m, t)
        stats = torch.zeros(n, parts, 2, dtype=torch.float64, device=x.device)
    if want_amax:
        amax = torch.zeros(n, lib().ps_conv1x1_stats_parts(m, t), dtype=torch.float32, device=x.device)
    if x_amax is not None:
        require_device(x_amax, "conv1x1_f16x2 (x_amax)")
        if x_amax.dim() != 2 or x_amax.shape[0] != n or not x_amax.is_contiguous():
            raise ValueError(f"conv1x1_f16x2: x_amax must be a contiguous [N={n}, parts] tensor, got {tuple(x_amax.shape)}")
    rng = F16x2Range(int(w_exp), float(x_bound), ptr(x_amax), x_amax.shape[1] if x_amax is not None else 0, ptr(amax))
    if amax_map is not None:
        rng.amax_mul, rng.amax_add = float(amax_map[0]), float(amax_map[1])
    check(lib().ps_conv1x1_f16x2_f32(ptr(x), ptr(wt_planes), C.byref(rng), ptr(y), n, k, m, t, ldt,
                                     C.byref(pro) if pro is not None else None, ptr(bias), ptr(bias_n), ptr(res),
                                     ptr(stats), stream_ptr(x.device)), "ps_conv1x1_f16x2_f32")
    return y, stats, amax


FMAJOR_PAD = int(os.environ.get("PS_FMAJOR_PAD", "0"))   # floats between the M outputs of consecutive frames


def fmajor_ld(m: int) -> int:
    """Frame stride of the frame-major gate pre-activations: M (+ PS_FMAJOR_PAD floats; a 256-byte skew between frames was
    tried against a suspected 4 KiB channel stride and made no difference: 1.40 / 1.39 / 1.31 ms per launch at 64 / 0 / 32)."""
    return m + FMAJOR_PAD


def conv1x1_f16x2_fmajor_ok(n: int, k: int, m: int, t: int, ldt: int) -> bool:
    return bool(lib().ps_conv1x1_f16x2_fmajor_ok(n, k, m, t, ldt, fmajor_ld(m)))


def conv1x1_f16x2_fmajor(x: torch.Tensor, t: int, wt_planes: torch.Tensor, w_exp: int, m: int,
                         bias: Optional[torch.Tensor] = None, x_bound: float = 0.0,
                         x_amax: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ps_conv1x1_f16x2_fmajor_f32: the fp16x2 GEMM writing y FRAME-MAJOR, a [N, ldt, M] view of rows fmajor_ld(M) apart
    (the gate pre-activations of lstm_fmajor).  Raises when the launch does not qualify (conv1x1_f16x2_fmajor_ok)."""
    require_device(x, "conv1x1_f16x2_fmajor")
    n, k, ldt = x.shape
    ldm = fmajor_ld(m)
    y = torch.empty(n, ldt, ldm, dtype=torch.float32, device=x.device)[..., :m]
    rng = F16x2Range(int(w_exp), float(x_bound), ptr(x_amax), x_amax.shape[1] if x_amax is not None else 0, None)
    check(lib().ps_conv1x1_f16x2_fmajor_f32(ptr(x), ptr(wt_planes), C.byref(rng), ptr(y), n, k, m, t, ldt, ldm, ptr(bias),
                                            stream_ptr(x.device)), "ps_conv1x1_f16x2_fmajor_f32")
    return y


def conv1x1_f16x2_ln_ok(n: int, k: int, c: int, t: int) -> bool:
    return bool(lib().ps_conv1x1_f16x2_ln_ok(n, k, c, t))


def conv1x1_f16x2_ln(x: torch.Tensor, t: int, wt_planes: torch.Tensor, w_exp: int, c: int, bias: Optional[torch.Tensor],
                     gamma: torch.Tensor, beta: torch.Tensor, eps: float, res: Optional[torch.Tensor],
                     x_bound: float = 0.0, x_amax: Optional[torch.Tensor] = None,
                     out: Optional[torch.Tensor] = None, want_amax: bool = False, pro: Optional[Prologue] = None,
                     res_inside: bool = False):
    """ps_conv1x1_f16x2_ln_f32: y = res + LayerNorm_channels(W x + b) in one launch (C = 128; wt_planes = pack_wt_f16x2 of
    W zero padded to 256 rows), or y = LayerNorm(W x + b + res) with res_inside; `pro`: affine / PReLU prologue on x.
    Raises when the launch does not qualify (conv1x1_f16x2_ln_ok).  want_amax: returns (y, [N, parts] partial maxima of
    |y|) -- the x_amax of the next fp16x2 GEMM."""
    require_device(x, "conv1x1_f16x2_ln")
    n, k, ldt = x.shape
    y = out if out is not None else torch.empty(n, c, ldt, dtype=torch.float32, device=x.device)
    if res is not None and (tuple(res.shape) != (n, c, ldt) or not res.is_contiguous()):
        raise ValueError(f"conv1x1_f16x2_ln: the residual must be a contiguous {(n, c, ldt)} tensor")
    amax = torch.zeros(n, lib().ps_conv1x1_stats_parts(256, t), dtype=torch.float32, device=x.device) if want_amax else None
    rng = F16x2Range(int(w_exp), float(x_bound), ptr(x_amax), x_amax.shape[1] if x_amax is not None else 0, ptr(amax))
    check(lib().ps_conv1x1_f16x2_ln_f32(ptr(x), ptr(wt_planes), C.byref(rng), ptr(y), n, k, c, t, ldt,
                                        C.byref(pro) if pro is not None else None, ptr(bias), ptr(gamma), ptr(beta), float(eps),
                                        ptr(res), int(res_inside), stream_ptr(x.device)), "ps_conv1x1_f16x2_ln_f32")
    return (y, amax) if want_amax else y


def absmax(x: torch.Tensor, t: int) -> torch.Tensor:
    """padded rows [N, C, ldt] -> [N, ps_absmax_parts()] partial maxima of |x| over the t valid frames."""
    require_device(x, "absmax")
    n, c, ldt = x.shape
    out = torch.empty(n, lib().ps_absmax_parts(), dtype=torch.float32, device=x.device)
    check(lib().ps_absmax_f32(ptr(x), ptr(out), n, c, t, ldt, stream_ptr(x.device)), "ps_absmax_f32")
    return out


def conv1x1_bf16(x: torch.Tensor, t: int, wt_planes: torch.Tensor, m: int, pro: Optional[Prologue] = None,
                 bias: Optional[torch.Tensor] = None, bias_n: Optional[torch.Tensor] = None,
                 res: Optional[torch.Tensor] = None, want_stats: bool = False,
                 out: Optional[torch.Tensor] = None,
                 out_dtype: torch.dtype = torch.float32) -> tuple[torch.Tensor, Optional[torch.Tensor]]:
    """ps_conv1x1_f32's contract on the bf16 matrix pipe; the plane count is read off wt_planes (pack_wt_bf16).
    A torch.bfloat16 `x` / `out` (or out_dtype) selects bf16 activation rows (ps_conv1x1_bf16_io, planes = 1)."""
    require_device(x, "conv1x1_bf16", allow_bf16=True)
    n, k, ldt = x.shape
    planes = wt_planes.shape[2]
    y = out if out is not None else torch.empty(n, m, ldt, dtype=out_dtype, device=x.device)
    if res is not None and res.dtype != y.dtype:
        raise ValueError(f"conv1x1_bf16: the residual rows must have the output's dtype ({y.dtype}), got {res.dtype}")
    stats = None
    if want_stats:
        parts = lib().ps_conv1x1_stats_parts(m, t)
        stats = torch.zeros(n, parts, 2, dtype=torch.float64, device=x.device)
    check(lib().ps_conv1x1_bf16_io(ptr(x), int(x.dtype == torch.bfloat16), ptr(wt_planes), ptr(y),
                                   int(y.dtype == torch.bfloat16), n, k, m, t, ldt, planes,
                                   C.byref(pro) if pro is not None else None, ptr(bias), ptr(bias_n), ptr(res),
                                   ptr(stats), stream_ptr(x.device)), "ps_conv1x1_bf16_io")
    return y, stats


def dwconv(x: torch.Tensor, t: int, w: torch.Tensor, b: Optional[torch.Tensor], dilation: int, left: int,
           pro: Optional[Prologue] = None, want_stats: bool = False,
           out_dtype: Optional[torch.dtype] = None, want_amax: bool = False) -> tuple[torch.Tensor, Optional[torch.Tensor]]:
    """x padded [N,H,ldt] (fp32 or bf16 rows), w [H,1,P] -> y padded [N,H,ldt] (+ partial stats; with want_amax the
    partial maxima of |y| [N, parts] instead: ps_dwconv_amax_f32, fp32 rows, P = 3, 2 * dilation <= 256)."""
    require_device(x, "dwconv", allow_bf16=True)
    n, h, ldt = x.shape
    p = w.shape[-1]
    y = torch.zeros(n, h, ldt, dtype=out_dtype or x.dtype, device=x.device)
    if want_amax:
        amax = torch.empty(n, lib().ps_dwconv_stats_parts(h, t), dtype=torch.float32, device=x.device)
        check(lib().ps_dwconv_amax_f32(ptr(x), ptr(w), ptr(b), ptr(y), n, h, t, ldt, p, dilation, left,
                                       C.byref(pro) if pro is not None else None, ptr(amax), stream_ptr(x.device)),
              "ps_dwconv_amax_f32")
        return y, amax
    stats = None
    if want_stats:
        parts = lib().ps_dwconv_stats_parts(h, t)
        stats = torch.zeros(n, parts, 2, dtype=torch.float64, device=x.device)
    check(lib().ps_dwconv_io(ptr(x), int(x.dtype == torch.bfloat16), ptr(w), ptr(b), ptr(y),
                             int(y.dtype == torch.bfloat16), n, h, t, ldt, p, dilation, left,
                             C.byref(pro) if pro is not None else None, ptr(stats), stream_ptr(x.device)),
          "ps_dwconv_io")
    return y, stats


def attn_stats_pool(logits: torch.Tensor, x: torch.Tensor, t: int, eps: float = 1e-12,
                    lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
    """attention logits / features padded [N,C,ldt] -> [N,2C] = cat(weighted mean, weighted std); `lengths` [N]:
    relative lengths, frames t with float(t) >= lengths[n] * T are masked out (lobe/pooling.py:100-107)."""
    require_device(x, "attn_stats_pool")
    n, c, ldt = x.shape
    out = torch.empty(n, 2 * c, dtype=torch.float32, device=x.device)
    if lengths is not None:
        require_device(lengths, "attn_stats_pool")
        if lengths.shape != (n,):
            raise RuntimeError("attn_stats_pool: lengths must be [N]")
        lengths = lengths.to(torch.float32).contiguous()
    check(lib().ps_attn_stats_pool_len_f32(ptr(logits), ptr(x), ptr(lengths), ptr(out), n, c, t, ldt, float(eps),
                                           stream_ptr(x.device)), "ps_attn_stats_pool_len_f32")
    return out


def attn_weights(logits: torch.Tensor, t: int, lengths: Optional[torch.Tensor] = None) -> torch.Tensor:
    """padded attention logits [N,C,ldt] (+ relative lengths [N]) -> padded softmax over the valid frames."""
    require_device(logits, "attn_weights")
    n, c, ldt = logits.shape
    out = torch.empty_like(logits)
    if lengths is not None:
        lengths = lengths.detach().to(device=logits.device, dtype=torch.float32).contiguous()
        if lengths.numel() != n:
            raise RuntimeError(f"attn_weights: lengths must hold one value per utterance ({n}), got {tuple(lengths.shape)}")
    check(lib().ps_attn_weights_f32(ptr(logits), ptr(lengths), ptr(out), n, c, t, ldt, stream_ptr(logits.device)),
          "ps_attn_weights_f32")
    return out


def lstm(gx: torch.Tensor, whh_t: torch.Tensor, hidden: int, dirs: int, q: int, q_stride: int, steps: int,
         step_stride: int, h0: Optional[torch.Tensor] = None, c0: Optional[torch.Tensor] = None,
         want_state: bool = False, state_shift: int = 0, state_out: Optional[tuple] = None, f16x2: bool = False,
         out: Optional[torch.Tensor] = None):
    """LSTM recurrence over gate pre-activations gx padded [N,D*4H,ldt] -> hout [N,D*H,ldt]
    (+ final (h, c) in state layout [N,D*H,ldq] when want_state / state_out).  f16x2: ps_lstm_f16x2_f32 (the
    recurrent product in two fp16 terms per operand where a kernel for it exists, else the fp32 kernels)."""
    require_device(gx, "lstm")
    n, rows, ldt = gx.shape
    if rows != dirs * 4 * hidden or tuple(whh_t.shape) != (dirs, hidden, 4 * hidden):
        raise RuntimeError("lstm: gx must be [N, D*4H, ldt] and whh_t [D, H, 4H]")
    hout = out if out is not None else torch.empty(n, dirs * hidden, ldt, dtype=torch.float32, device=gx.device)
    if tuple(hout.shape) != (n, dirs * hidden, ldt) or not hout.is_contiguous():
        raise RuntimeError("lstm: out must be a contiguous [N, D*H, ldt] tensor")
    a = LstmArgs()
    a.gx, a.whh_t, a.hout = ptr(gx), ptr(whh_t), ptr(hout)
    ldq = padded_frames(q)
    for name, t in (("h0", h0), ("c0", c0)):
        if t is not None and (tuple(t.shape[:2]) != (n, dirs * hidden) or t.shape[2] < q or not t.is_contiguous()):
            raise RuntimeError(f"lstm: {name} must be a contiguous state tensor [N, D*H, ldq >= Q]")
    if h0 is not None:
        ldq = h0.shape[2]
    elif c0 is not None:
        ldq = c0.shape[2]
    if (h0 is not None and c0 is not None) and h0.shape[2] != c0.shape[2]:
        raise RuntimeError("lstm: h0 and c0 must share ldq")
    h_last = c_last = None
    if state_out is not None:
        h_last, c_last = state_out
        if h_last.shape[2] != ldq and (h0 is not None or c0 is not None):
            raise RuntimeError("lstm: state_out must share ldq with h0/c0")
        ldq = h_last.shape[2]
    elif want_state:
        h_last = torch.zeros(n, dirs * hidden, ldq, dtype=torch.float32, device=gx.device)
        c_last = torch.zeros_like(h_last)
    a.h0, a.c0, a.h_last, a.c_last = ptr(h0), ptr(c0), ptr(h_last), ptr(c_last)
    a.N, a.H, a.D, a.Q, a.q_stride, a.steps, a.step_stride = n, hidden, dirs, q, q_stride, steps, step_stride
    a.ldt, a.ldq, a.state_shift = ldt, ldq, state_shift
    if f16x2:
        check(lib().ps_lstm_f16x2_f32(C.byref(a), stream_ptr(gx.device)), "ps_lstm_f16x2_f32")
    else:
        check(lib().ps_lstm_f32(C.byref(a), stream_ptr(gx.device)), "ps_lstm_f32")
    return (hout, (h_last, c_last)) if h_last is not None else (hout, None)


RNN_KINDS = {"RNN": 0, "GRU": 2}


def rnn(gx: torch.Tensor, whh_t: torch.Tensor, kind: str, hidden: int, dirs: int, q: int, q_stride: int, steps: int,
        step_stride: int, bhn: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ps_rnn_f32: the recurrence of nn.RNN (tanh) / nn.GRU over pre-activations gx padded [N, D*G, ldt] (G = H or 3H) ->
    hout [N, D*H, ldt]; zero initial state."""
    require_device(gx, "rnn")
    n, rows, ldt = gx.shape
    g = (3 if kind == "GRU" else 1) * hidden
    if kind not in RNN_KINDS or rows != dirs * g or tuple(whh_t.shape) != (dirs, hidden, g):
        raise RuntimeError("rnn: kind RNN / GRU, gx [N, D*G, ldt], whh_t [D, H, G]")
    hout = torch.empty(n, dirs * hidden, ldt, dtype=torch.float32, device=gx.device)
    a = LstmArgs()
    a.gx, a.whh_t, a.hout = ptr(gx), ptr(whh_t), ptr(hout)
    a.N, a.H, a.D, a.Q, a.q_stride, a.steps, a.step_stride = n, hidden, dirs, q, q_stride, steps, step_stride
    a.ldt, a.ldq, a.state_shift = ldt, 0, 0
    check(lib().ps_rnn_f32(C.byref(a), RNN_KINDS[kind], ptr(bhn), stream_ptr(gx.device)), "ps_rnn_f32")
    return hout


def _lstm_fmajor_args(gx_fm: torch.Tensor, whh_t: torch.Tensor, hout, hidden, dirs, q, q_stride, steps, step_stride):
    n, ldt, rows = gx_fm.shape
    a = LstmArgs()
    a.gx, a.whh_t, a.hout = ptr(gx_fm), ptr(whh_t), ptr(hout)
    a.N, a.H, a.D, a.Q, a.q_stride, a.steps, a.step_stride = n, hidden, dirs, q, q_stride, steps, step_stride
    a.ldt, a.ldq, a.state_shift = ldt, 0, 0
    return a


def lstm_fmajor_ok(n: int, ldt: int, hidden: int, dirs: int, q: int, q_stride: int, steps: int, step_stride: int) -> bool:
    """Does ps_lstm_fmajor_f16x2_f32 take this pass?  (shape / stride conditions only: H = 128, no states, slabs < 2 GiB)"""
    if hidden != 128 or dirs not in (1, 2) or (q - 1) * q_stride + (steps - 1) * step_stride >= ldt:
        return False
    return ldt * fmajor_ld(dirs * 512) * 4 < 2 ** 31


def lstm_fmajor(gx_fm: torch.Tensor, whh_t: torch.Tensor, hidden: int, dirs: int, q: int, q_stride: int, steps: int,
                step_stride: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ps_lstm_fmajor_f16x2_f32: LSTM recurrence over FRAME-MAJOR gate pre-activations gx [N, ldt, D*4H] (from
    conv1x1_f16x2_fmajor; the frames may be padded rows: stride(1) >= D*4H) -> hout [N, D*H, ldt]; H = 128, zero initial
    states, fp16x2 recurrent product."""
    require_device(gx_fm, "lstm_fmajor")
    n, ldt, rows = gx_fm.shape
    if rows != dirs * 4 * hidden or tuple(whh_t.shape) != (dirs, hidden, 4 * hidden):
        raise RuntimeError("lstm_fmajor: gx must be [N, ldt, D*4H] and whh_t [D, H, 4H]")
    ldm = gx_fm.stride(1)
    if gx_fm.stride(2) != 1 or gx_fm.stride(0) != ldt * ldm:
        raise RuntimeError("lstm_fmajor: gx must be a [N, ldt, :D*4H] view of contiguous frame rows")
    hout = out if out is not None else torch.empty(n, dirs * hidden, ldt, dtype=torch.float32, device=gx_fm.device)
    if tuple(hout.shape) != (n, dirs * hidden, ldt) or not hout.is_contiguous():
        raise RuntimeError("lstm_fmajor: out must be a contiguous [N, D*H, ldt] tensor")
    a = _lstm_fmajor_args(gx_fm, whh_t, hout, hidden, dirs, q, q_stride, steps, step_stride)
    check(lib().ps_lstm_fmajor_f16x2_f32(C.byref(a), ldm, stream_ptr(gx_fm.device)), "ps_lstm_fmajor_f16x2_f32")
    return hout


def pack_whh_h256(whh_t: torch.Tensor):
    """weight_hh^T [D, H, 4H], H = 256 or 192 -> (fp16 image [D, H/32, H/32, 2, 4, 2, 64, 8], acc scales [D] as a Python list)
    for lstm_fmajor_h256 (layout: include/puresound_hip.h).  Reads the maxima back to the host: plan-build time only."""
    import math
    d, hh, g = whh_t.shape
    if hh not in (256, 192) or g != 4 * hh:
        raise ValueError("pack_whh_h256: whh_t must be [D, H, 4H] with H = 256 or 192")
    nw = hh // 32
    imgs, scales = [], []
    for i in range(d):
        w = whh_t[i].detach().float().t().contiguous()            # [4H gate rows, H k]
        wmax = float(w.abs().max())
        if not (wmax < float("inf")):
            raise ValueError("pack_whh_h256: the weight holds inf / NaN")
        ex = max(math.frexp(wmax)[1], -27) if wmax > 0 else 0
        sc = math.ldexp(1.0, 13 - ex)
        rest = w * sc
        planes = []
        for _ in range(2):
            hp = rest.to(torch.float16)
            planes.append(hp)
            rest = rest - hp.float()
        pl = torch.stack(planes, 0)                                 # [pl, 4H, H]
        pl = pl.reshape(2, 4, nw, 2, 16, nw, 4, 8)                  # pl, g, w, rb, row, ks, kg, e
        pl = pl.permute(2, 5, 3, 1, 0, 6, 4, 7)                     # w, ks, rb, g, pl, kg, row, e
        imgs.append(pl.reshape(nw, nw, 2, 4, 2, 64, 8))
        scales.append(sc * 1024.0)
    return torch.stack(imgs, 0).contiguous(), scales


def lstm_fmajor_h256_ok(n: int, ldt: int, dirs: int, q: int, q_stride: int, steps: int, step_stride: int) -> bool:
    return dirs in (1, 2) and (q - 1) * q_stride + (steps - 1) * step_stride < ldt


def lstm_fmajor_h256(gx_fm: torch.Tensor, whh_image: torch.Tensor, acc_scale, dirs: int, q: int, q_stride: int, steps: int,
                     step_stride: int, h0: Optional[torch.Tensor] = None, c0: Optional[torch.Tensor] = None,
                     want_state: bool = False, state_shift: int = 0, state_out: Optional[tuple] = None,
                     out: Optional[torch.Tensor] = None):
    """ps_lstm_fmajor_h256_f16x2_f32: the H = 256 recurrence over FRAME-MAJOR pre-activations gx [N, ldt, D*1024] with W_hh
    streamed from its packed image (pack_whh_h256) -> (hout [N, D*256, ldt], final (h, c) or None), states as lstm()."""
    require_device(gx_fm, "lstm_fmajor_h256")
    hidden = whh_image.shape[1] * 32
    n, ldt, rows = gx_fm.shape
    if (rows != dirs * 4 * hidden or hidden not in (256, 192) or whh_image.dtype != torch.float16
            or tuple(whh_image.shape) != (dirs, hidden // 32, hidden // 32, 2, 4, 2, 64, 8)):
        raise RuntimeError("lstm_fmajor_h256: gx must be [N, ldt, D*4H] and the image pack_whh_h256's")
    ldm = gx_fm.stride(1)
    if gx_fm.stride(2) != 1 or gx_fm.stride(0) != ldt * ldm:
        raise RuntimeError("lstm_fmajor_h256: gx must be a [N, ldt, :D*1024] view of contiguous frame rows")
    hout = out if out is not None else torch.empty(n, dirs * hidden, ldt, dtype=torch.float32, device=gx_fm.device)
    a = LstmArgs()
    a.gx, a.whh_t, a.hout = ptr(gx_fm), None, ptr(hout)
    ldq = padded_frames(q)
    for name, t in (("h0", h0), ("c0", c0)):
        if t is not None and (tuple(t.shape[:2]) != (n, dirs * hidden) or t.shape[2] < q or not t.is_contiguous()):
            raise RuntimeError(f"lstm_fmajor_h256: {name} must be a contiguous state tensor [N, D*H, ldq >= Q]")
    if h0 is not None:
        ldq = h0.shape[2]
    elif c0 is not None:
        ldq = c0.shape[2]
    h_last = c_last = None
    if state_out is not None:
        h_last, c_last = state_out
        ldq = h_last.shape[2]
    elif want_state:
        h_last = torch.zeros(n, dirs * hidden, ldq, dtype=torch.float32, device=gx_fm.device)
        c_last = torch.zeros_like(h_last)
    a.h0, a.c0, a.h_last, a.c_last = ptr(h0), ptr(c0), ptr(h_last), ptr(c_last)
    a.N, a.H, a.D, a.Q, a.q_stride, a.steps, a.step_stride = n, hidden, dirs, q, q_stride, steps, step_stride
    a.ldt, a.ldq, a.state_shift = ldt, ldq, state_shift
    sc = (C.c_float * 2)(float(acc_scale[0]), float(acc_scale[-1]))
    # few sequence groups (a speaker LSTM over all frames, SkiM's Mem-LSTMs): the cooperative kernel -- W_hh resident in the
    # registers of H / 32 CUs per group -- instead of streaming 1 MiB of it through one CU every step
    ws_bytes = lib().ps_lstm_fmajor_coop_workspace_bytes(C.byref(a), ldm) if COOP_LSTM else 0
    if ws_bytes:
        ws = torch.empty(ws_bytes, dtype=torch.uint8, device=gx_fm.device)
        check(lib().ps_lstm_fmajor_coop_f16x2_f32(C.byref(a), ldm, ptr(whh_image), sc, ptr(ws), ws_bytes,
                                                  stream_ptr(gx_fm.device)), "ps_lstm_fmajor_coop_f16x2_f32")
        _COOP_LAST[0] = ws   # (tests read the error word behind the counters: its last four bytes' block)
    else:
        check(lib().ps_lstm_fmajor_h256_f16x2_f32(C.byref(a), ldm, ptr(whh_image), sc, stream_ptr(gx_fm.device)),
              "ps_lstm_fmajor_h256_f16x2_f32")
    return (hout, (h_last, c_last)) if h_last is not None else (hout, None)


COOP_LSTM = os.environ.get("PS_COOP_LSTM", "1") != "0"   # 0: always the streamed-weight kernel
_COOP_LAST = [None]


def _coop_slices(dirs: int, groups: int, hidden: int) -> int:
    """slices per group the launcher picks: H / 32 (two waves each) while the launch fits the chip, else H / 64"""
    rounds = (groups * dirs + 7) // 8 * 8   # (both directions in one launch; one launch per direction beyond that: H / 64 too)
    cus = torch.cuda.get_device_properties(torch.cuda.current_device()).multi_processor_count
    return hidden // 32 if rounds * (hidden // 32) <= cus else hidden // 64


def coop_lstm_error_word(dirs: int, groups: int, hidden: int) -> int:
    """The error word of the last cooperative LSTM launch's workspace (1 = a group barrier gave up waiting); synchronises."""
    ws = _COOP_LAST[0]
    if ws is None:
        return 0
    hx = (2 * dirs * groups * 2 * 16 * (hidden + 8) * 2 + 255) // 256 * 256
    return int(ws.view(torch.int32)[hx // 4 + dirs * groups])


def coop_lstm_xcd_ids(dirs: int, groups: int, hidden: int) -> torch.Tensor:
    """[D * groups, slices]: the XCD every slice of the last cooperative launch ran on (the light group barrier needs each
    row constant: tests check it on the target part)."""
    ws = _COOP_LAST[0]
    hx = (2 * dirs * groups * 2 * 16 * (hidden + 8) * 2 + 255) // 256 * 256
    first = hx // 4 + dirs * groups + 1
    ns = _coop_slices(dirs, groups, hidden)
    return ws.view(torch.int32)[first:first + dirs * groups * ns].reshape(dirs * groups, ns).cpu()


def coop_lstm_xcd_masks(dirs: int, groups: int, hidden: int) -> torch.Tensor:
    """[D * groups]: per cluster the OR of 1 << XCD over its slices (one bit = the cluster took the light barrier)."""
    ws = _COOP_LAST[0]
    hx = (2 * dirs * groups * 2 * 16 * (hidden + 8) * 2 + 255) // 256 * 256
    first = hx // 4 + dirs * groups + 1 + dirs * groups * _coop_slices(dirs, groups, hidden)
    return ws.view(torch.int32)[first:first + dirs * groups].cpu()


def chan_layernorm(x: torch.Tensor, t: int, gamma: torch.Tensor, beta: torch.Tensor, eps: float,
                   res: Optional[torch.Tensor] = None, slope: Optional[torch.Tensor] = None, sigmoid: bool = False,
                   mul: Optional[torch.Tensor] = None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """[res +] [mul *] act(LN over channels of every frame) on padded [N,C,ldt]."""
    require_device(x, "chan_layernorm")
    n, c, ldt = x.shape
    y = out if out is not None else torch.empty_like(x)
    check(lib().ps_chan_layernorm_f32(ptr(x), ptr(gamma), ptr(beta), float(eps), ptr(slope), int(sigmoid), ptr(mul),
                                      ptr(res), ptr(y), n, c, t, ldt, stream_ptr(x.device)), "ps_chan_layernorm_f32")
    return y


def unfold_taps(x: torch.Tensor, t: int, taps: int, dilation: int, left: int, scale: Optional[torch.Tensor] = None,
                shift: Optional[torch.Tensor] = None, embed: Optional[torch.Tensor] = None,
                t_out: Optional[int] = None) -> torch.Tensor:
    """x padded [N,K,ldt] -> [N, taps*(K+E), ldt]: tap-shifted copies (zero outside [0,T)), optional per-(n,k) affine
    before the padding, optional constant embedding rows.  t_out > t: that many output frames per row."""
    require_device(x, "unfold_taps")
    n, k, ldt = x.shape
    e = 0 if embed is None else embed.shape[1]
    t_out = t if t_out is None else t_out
    y = torch.empty(n, taps * (k + e), ldt, dtype=torch.float32, device=x.device)
    check(lib().ps_unfold_taps_out_f32(ptr(x), ptr(y), n, k, t, t_out, ldt, taps, dilation, left, ptr(scale), ptr(shift),
                                       ptr(embed), e, stream_ptr(x.device)), "ps_unfold_taps_out_f32")
    return y


def gated_product(left: torch.Tensor, right: torch.Tensor, t: int, pro_left: Prologue, pro_right: Prologue) -> torch.Tensor:
    """PReLU(norm(left)) * sigmoid(PReLU(norm(right))) on padded [N,H,ldt]."""
    require_device(left, "gated_product")
    n, h, ldt = left.shape
    y = torch.empty_like(left)
    check(lib().ps_gated_product_f32(ptr(left), ptr(right), ptr(y), n, h, t, ldt, C.byref(pro_left), C.byref(pro_right),
                                     stream_ptr(left.device)), "ps_gated_product_f32")
    return y


def overlap_geometry(t: int, k: int) -> tuple[int, int]:
    """(rest, S*K) of the 50 % overlapped segmentation of t frames into k-frame segments."""
    stride = k // 2
    rest = k - (stride + t % k) % k
    return rest, 2 * ((t + rest + stride) // k) * k


def segment_split(x: torch.Tensor, t: int, k: int) -> tuple[torch.Tensor, int]:
    """padded [N,C,ldt] with t frames -> (padded [N,C,ld'] holding S*K frames of overlapped segments, S*K)."""
    require_device(x, "segment_split")
    n, c, ldt = x.shape
    _, tp = overlap_geometry(t, k)
    y = torch.empty(n, c, padded_frames(tp), dtype=torch.float32, device=x.device)
    check(lib().ps_segment_overlap_f32(ptr(x), ptr(y), n * c, t, ldt, tp, y.shape[-1], k, 0, stream_ptr(x.device)),
          "ps_segment_overlap_f32")
    return y, tp


def segment_merge(x: torch.Tensor, tp: int, t: int, k: int) -> torch.Tensor:
    """inverse of segment_split: padded [N,C,ld'] with tp segment frames -> padded [N,C,ldt] with t frames."""
    require_device(x, "segment_merge")
    n, c, ld = x.shape
    y = torch.empty(n, c, padded_frames(t), dtype=torch.float32, device=x.device)
    check(lib().ps_segment_overlap_f32(ptr(x), ptr(y), n * c, tp, ld, t, y.shape[-1], k, 1, stream_ptr(x.device)),
          "ps_segment_overlap_f32")
    return y


def film_conv(x: torch.Tensor, t: int, wt_pairs: torch.Tensor, res_pairs: Optional[torch.Tensor],
              out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """FiLM (after its input norm) in one kernel: (Ws x + rs) * x + (Wb x + rb); weight rows paired (scale, bias)."""
    require_device(x, "film_conv")
    n, c, ldt = x.shape
    y = out if out is not None else torch.empty_like(x)
    check(lib().ps_film_conv_f32(ptr(x), ptr(wt_pairs), ptr(res_pairs), ptr(y), n, c, t, ldt, stream_ptr(x.device)),
          "ps_film_conv_f32")
    return y


def lstm_gates_cell(xh: torch.Tensor, t: int, wt_units: torch.Tensor, bias_units: torch.Tensor, c: torch.Tensor,
                    h: torch.Tensor, hidden: int) -> None:
    """gates GEMM over [x; h] + LSTM cell; c in place, h' into `h` (not the h rows of xh)."""
    require_device(xh, "lstm_gates_cell")
    n, k, ldt = xh.shape
    if n != 1 and (not c.is_contiguous() or not h.is_contiguous()):
        raise RuntimeError("lstm_gates_cell: state views are supported for N = 1 only")
    check(lib().ps_lstm_gates_cell_f32(ptr(xh), ptr(wt_units), ptr(bias_units), ptr(c), ptr(h), n, k, hidden, t, ldt,
                                       c.stride(1), stream_ptr(xh.device)), "ps_lstm_gates_cell_f32")


def proj_layernorm(x: torch.Tensor, t: int, wt: torch.Tensor, bias: Optional[torch.Tensor], m: int,
                   gamma: torch.Tensor, beta: torch.Tensor, eps: float, res: Optional[torch.Tensor],
                   norm2: Optional[tuple] = None, x_copy: Optional[torch.Tensor] = None, res_inside: bool = False,
                   out: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None, want_amax: bool = False):
    """y = res + LN(W x + b), or LN(W x + b + res) with res_inside (+ y2 = LN2(y), + copy of x); returns (y, y2);
    y is written into `out` (contiguous [N, m, ldt]) when given."""
    require_device(x, "proj_layernorm")
    n, k, ldt = x.shape
    if out is not None and (tuple(out.shape) != (n, m, ldt) or not out.is_contiguous() or out.dtype != torch.float32):
        raise ValueError(f"proj_layernorm: `out` must be a contiguous fp32 {(n, m, ldt)} tensor")
    y = out if out is not None else torch.empty(n, m, ldt, dtype=torch.float32, device=x.device)
    y2 = (out2 if out2 is not None else torch.empty_like(y)) if norm2 is not None else None
    g2, b2, e2 = norm2 if norm2 is not None else (None, None, 0.0)
    if want_amax:  # (long rows only: the row kernel; the caller falls back to absmax() when this raises PS_E_UNSUPPORTED)
        amax = torch.empty(n, lib().ps_proj_layernorm_amax_parts(t), dtype=torch.float32, device=x.device)
        check(lib().ps_proj_layernorm_amax_f32(ptr(x), ptr(wt), ptr(bias), ptr(gamma), ptr(beta), float(eps), ptr(res), ptr(y),
                                               ptr(g2), ptr(b2), float(e2), ptr(y2), ptr(x_copy), int(res_inside), n, k, m, t,
                                               ldt, ptr(amax), stream_ptr(x.device)), "ps_proj_layernorm_amax_f32")
        return y, y2, amax
    check(lib().ps_proj_layernorm_f32(ptr(x), ptr(wt), ptr(bias), ptr(gamma), ptr(beta), float(eps), ptr(res), ptr(y),
                                      ptr(g2), ptr(b2), float(e2), ptr(y2), ptr(x_copy), int(res_inside), n, k, m, t, ldt,
                                      stream_ptr(x.device)), "ps_proj_layernorm_f32")
    return y, y2


# ---- the cells of a streaming anti-diagonal: the three operators above for several one-utterance problems per launch ----
MAX_CELLS = _abi.PS_MAX_CELLS


def film_conv_cells(cells: list, t: int) -> None:
    """cells: [(xn [1,C,ld], wt_pairs, res_pairs | None, out [1,C,ld]), ...] -> one ps_film_conv_cells_f32 launch."""
    x0 = cells[0][0]
    require_device(x0, "film_conv_cells")
    _, c, ldt = x0.shape
    arr = (_abi.FilmCell * len(cells))()
    for a, (x, wt, res, y) in zip(arr, cells):
        if tuple(x.shape) != (1, c, ldt) or tuple(y.shape) != (1, c, ldt) or x.stride(1) != ldt or y.stride(1) != ldt:
            raise RuntimeError("film_conv_cells: every cell is a [1, C, ld] row block of one shape")
        a.x, a.wt_pairs, a.res_pairs, a.y = ptr(x), ptr(wt), ptr(res), ptr(y)
    check(lib().ps_film_conv_cells_f32(arr, len(cells), c, t, ldt, stream_ptr(x0.device)), "ps_film_conv_cells_f32")


def lstm_gates_cell_cells(cells: list, t: int, hidden: int) -> None:
    """cells: [(xh [1,K,ld], wt_units, bias_units, c [1,H,ld'], h [1,H,ld']), ...] -> one launch (c in place, h' out)."""
    xh0, c0 = cells[0][0], cells[0][3]
    require_device(xh0, "lstm_gates_cell_cells")
    _, k, ldt = xh0.shape
    arr = (_abi.GatesCell * len(cells))()
    for a, (xh, wt, bias, c, h) in zip(arr, cells):
        if tuple(xh.shape) != (1, k, ldt) or c.stride(1) != c0.stride(1) or h.stride(1) != c0.stride(1):
            raise RuntimeError("lstm_gates_cell_cells: every cell has the same [1, K, ld] block and state row stride")
        a.xh, a.wt_units, a.bias_units, a.c, a.h = ptr(xh), ptr(wt), ptr(bias), ptr(c), ptr(h)
    check(lib().ps_lstm_gates_cell_cells_f32(arr, len(cells), k, hidden, t, ldt, c0.stride(1), stream_ptr(xh0.device)),
          "ps_lstm_gates_cell_cells_f32")


def proj_layernorm_cells(cells: list, t: int, m: int) -> None:
    """cells: [dict(x, wt, bias, gamma, beta, eps, res, y, norm2 = (gamma2, beta2, eps2) | None, y2, x_copy), ...] -> one
    launch: y = res + LN(W x + b) (+ y2 = LN2(y), + x_copy = x) per cell."""
    x0 = cells[0]["x"]
    require_device(x0, "proj_layernorm_cells")
    _, k, ldt = x0.shape
    arr = (_abi.ProjLnCell * len(cells))()
    for a, c in zip(arr, cells):
        if tuple(c["x"].shape) != (1, k, ldt) or tuple(c["y"].shape) != (1, m, ldt) or not c["y"].is_contiguous():
            raise RuntimeError("proj_layernorm_cells: every cell maps a [1, K, ld] block to a contiguous [1, M, ld] block")
        g2, b2, e2 = c["norm2"] if c.get("norm2") is not None else (None, None, 0.0)
        a.x, a.wt, a.bias, a.gamma, a.beta = ptr(c["x"]), ptr(c["wt"]), ptr(c.get("bias")), ptr(c["gamma"]), ptr(c["beta"])
        a.res, a.y, a.gamma2, a.beta2 = ptr(c.get("res")), ptr(c["y"]), ptr(g2), ptr(b2)
        a.y2, a.x_copy = ptr(c.get("y2") if g2 is not None else None), ptr(c.get("x_copy"))
        a.eps, a.eps2 = float(c["eps"]), float(e2)
    check(lib().ps_proj_layernorm_cells_f32(arr, len(cells), 0, k, m, t, ldt, stream_ptr(x0.device)),
          "ps_proj_layernorm_cells_f32")


def overlap_average(prev: torch.Tensor, cur: torch.Tensor, overlap: int) -> torch.Tensor:
    """Streaming harness OLA: returns the new [B, win] block whose first `overlap` samples are the average of the
    tail of `prev` [B, L] and the head of `cur` [B, win]."""
    require_device(cur, "overlap_average")
    b, win = cur.shape
    if prev.shape[0] != b or prev.shape[1] < overlap or prev.stride(1) != 1:
        raise RuntimeError("overlap_average: prev must be [B, L >= overlap] with contiguous rows")
    cur = cur.contiguous()
    out = torch.empty_like(cur)
    tail = prev[:, prev.shape[1] - overlap:]
    check(lib().ps_overlap_average_f32(ptr(tail), prev.stride(0), ptr(cur), ptr(out), b, win, overlap,
                                       stream_ptr(cur.device)), "ps_overlap_average_f32")
    return out


def stream_windows(queue: torch.Tensor, chunk: torch.Tensor, wins: torch.Tensor, hop: int) -> None:
    """queue [B, 2 hop] ‖ chunk [B, hops * hop] -> wins [hops, B * 2 hop]: window i of stream b (ps_stream_windows_f32)."""
    require_device(chunk, "stream_windows")
    b, win = queue.shape
    hops = chunk.shape[1] // hop
    if not (queue.is_contiguous() and chunk.is_contiguous() and wins.is_contiguous()) or wins.shape != (hops, b * win):
        raise RuntimeError("stream_windows: contiguous queue [B, win], chunk [B, hops * hop], wins [hops, B * win] expected")
    check(lib().ps_stream_windows_f32(ptr(queue), ptr(chunk), ptr(wins), b, hops, win, hop, stream_ptr(chunk.device)),
          "ps_stream_windows_f32")


def stream_overlap(frames: torch.Tensor, wins: torch.Tensor, tail: torch.Tensor, blocks: torch.Tensor,
                   queue: torch.Tensor, hop: int) -> None:
    """Averaging overlap-add of all hops of a chunk, in place on tail / blocks / queue (ps_stream_overlap_f32)."""
    require_device(frames, "stream_overlap")
    b, win = queue.shape
    hops = wins.shape[0]
    if not all(t.is_contiguous() for t in (frames, wins, tail, blocks, queue)) or frames.numel() != hops * b * win \
            or tail.shape != (b, hop) or blocks.shape != (b, hops * hop):
        raise RuntimeError("stream_overlap: frames [hops, B, win], tail [B, hop], blocks [B, hops * hop] (contiguous) expected")
    check(lib().ps_stream_overlap_f32(ptr(frames), ptr(wins), ptr(tail), ptr(blocks), ptr(queue), b, hops, win, hop,
                                      stream_ptr(frames.device)), "ps_stream_overlap_f32")


ACT_KINDS = {"none": 0, "relu": 1, "prelu": 2, "mish": 3, "sigmoid": 4, "tanh": 5}


def unfold2d(x1: torch.Tensor, x2: Optional[torch.Tensor], t: int, f_out: int, kf: int, kt: int, stride_f: int,
             dil_f: int, dil_t: int, pad_f: int, pad_t: int, transposed: bool, t_in: Optional[int] = None) -> torch.Tensor:
    """x1 [N,C1,F,ld] (+ x2 [N,C2,F,ld]) -> tap rows [N, (C1+C2)*kf*kt, f_out*ld] for the Conv2d / ConvTranspose2d GEMM."""
    require_device(x1, "unfold2d")
    n, c1, f_in, ld = x1.shape
    c2 = 0 if x2 is None else x2.shape[1]
    if x2 is not None and (x2.shape[0], x2.shape[2], x2.shape[3]) != (n, f_in, ld):
        raise RuntimeError("unfold2d: the two sources must agree in N, F and ld")
    y = torch.empty(n, (c1 + c2) * kf * kt, f_out * ld, dtype=torch.float32, device=x1.device)
    check(lib().ps_unfold2d_f32(ptr(x1), c1, ptr(x2), c2, ptr(y), n, f_in, t if t_in is None else t_in, t, ld, kf, kt, stride_f, dil_f, dil_t, pad_f,
                                pad_t, f_out, int(transposed), stream_ptr(x1.device)), "ps_unfold2d_f32")
    return y


def conv2d(x1: torch.Tensor, x2: Optional[torch.Tensor], wt: torch.Tensor, bias: Optional[torch.Tensor], m: int, t: int,
           f_out: int, kf: int, kt: int, stride_f: int, dil_f: int, dil_t: int, pad_f: int, pad_t: int, transposed: bool,
           act: str = "none", slope: Optional[torch.Tensor] = None, t_in: Optional[int] = None) -> torch.Tensor:
    """Implicit-GEMM Conv2d / ConvTranspose2d on [N,C,F,ld] rows -> [N,M,f_out,ld] with bias + activation."""
    require_device(x1, "conv2d")
    n, c1, f_in, ld = x1.shape
    c2 = 0 if x2 is None else x2.shape[1]
    if x2 is not None and (x2.shape[0], x2.shape[2], x2.shape[3]) != (n, f_in, ld):
        raise RuntimeError("conv2d: the two sources must agree in N, F and ld")
    y = torch.empty(n, m, f_out, ld, dtype=torch.float32, device=x1.device)
    check(lib().ps_conv2d_f32(ptr(x1), c1, ptr(x2), c2, ptr(wt), ptr(bias), ptr(y), n, m, f_in,
                              t if t_in is None else t_in, t, ld, kf, kt, stride_f, dil_f, dil_t, pad_f, pad_t, f_out,
                              int(transposed), ACT_KINDS[act], ptr(slope), stream_ptr(x1.device)), "ps_conv2d_f32")
    return y


def pack_conv2d_f16x2(w2: torch.Tensor):
    """[M, K] fp32 (BatchNorm folded) -> (image of 2^w_exp W for ps_conv2d_f16x2_f32, w_exp); layout: include/puresound_hip.h.
    Reads the maximum back to the host: plan-build time only."""
    import math
    m, k = w2.shape
    wmax = float(w2.detach().abs().max())
    if not (wmax < float("inf")):
        raise ValueError("pack_conv2d_f16x2: the weight holds inf / NaN")
    w_exp = 13 - math.frexp(wmax)[1] + 1 if wmax > 0 else 0
    mt = 32 if m <= 32 else 64 if m <= 64 else 128
    tiles, nch = (m + mt - 1) // mt, (k + 31) // 32
    rest = torch.zeros(tiles * mt, nch * 32, dtype=torch.float32, device=w2.device)
    rest[:m, :k] = torch.ldexp(w2.detach().float(), torch.tensor(w_exp, device=w2.device))
    planes = []
    for _ in range(2):
        hp = rest.to(torch.float16)
        planes.append(hp)
        rest = rest - hp.float()
    img = torch.stack(planes, 0).reshape(2, tiles, mt // 16, 16, nch, 4, 8)      # pl, tile, rb, row, chunk, kg, e
    img = img.permute(1, 4, 0, 2, 5, 3, 6).contiguous()                          # tile, chunk, pl, rb, kg, row, e
    return img, w_exp


def conv2d_f16x2(x1: torch.Tensor, x2: Optional[torch.Tensor], wimg: torch.Tensor, w_exp: int, bias: Optional[torch.Tensor],
                 m: int, t: int, f_out: int, kf: int, kt: int, stride_f: int, dil_f: int, dil_t: int, pad_f: int, pad_t: int,
                 transposed: bool, act: str = "none", slope: Optional[torch.Tensor] = None, t_in: Optional[int] = None,
                 want_stats: bool = False):
    """conv2d / conv2d_stats in the fp16x2 arithmetic (ps_conv2d_f16x2_f32; weights from pack_conv2d_f16x2).  Returns y, or
    (y, stats) with want_stats (the activation should then be "none": a gLN follows)."""
    require_device(x1, "conv2d_f16x2")
    n, c1, f_in, ld = x1.shape
    c2 = 0 if x2 is None else x2.shape[1]
    if x2 is not None and (x2.shape[0], x2.shape[2], x2.shape[3]) != (n, f_in, ld):
        raise RuntimeError("conv2d_f16x2: the two sources must agree in N, F and ld")
    y = torch.empty(n, m, f_out, ld, dtype=torch.float32, device=x1.device)
    stats = None
    if want_stats:
        stats = torch.empty(n, lib().ps_conv2d_stats_parts(m, f_out, ld), 2, dtype=torch.float64, device=x1.device)
    check(lib().ps_conv2d_f16x2_f32(ptr(x1), c1, ptr(x2), c2, ptr(wimg), int(w_exp), ptr(bias), ptr(y), n, m, f_in,
                                    t if t_in is None else t_in, t, ld, kf, kt, stride_f, dil_f, dil_t, pad_f, pad_t, f_out,
                                    int(transposed), ACT_KINDS[act], ptr(slope), ptr(stats), stream_ptr(x1.device)),
          "ps_conv2d_f16x2_f32")
    return (y, stats) if want_stats else y


def conv2d_stats(x1: torch.Tensor, x2: Optional[torch.Tensor], wt: torch.Tensor, bias: Optional[torch.Tensor], m: int, t: int,
                 f_out: int, kf: int, kt: int, stride_f: int, dil_f: int, dil_t: int, pad_f: int, pad_t: int, transposed: bool,
                 t_in: Optional[int] = None):
    """conv2d without activation + the partial (sum, sum of squares) of its outputs over the t valid frames ->
    (y [N,M,f_out,ld], stats [N, parts, 2] fp64): the convolution in front of a gLN (ps_conv2d_stats_f32)."""
    require_device(x1, "conv2d_stats")
    n, c1, f_in, ld = x1.shape
    c2 = 0 if x2 is None else x2.shape[1]
    if x2 is not None and (x2.shape[0], x2.shape[2], x2.shape[3]) != (n, f_in, ld):
        raise RuntimeError("conv2d_stats: the two sources must agree in N, F and ld")
    y = torch.empty(n, m, f_out, ld, dtype=torch.float32, device=x1.device)
    stats = torch.empty(n, lib().ps_conv2d_stats_parts(m, f_out, ld), 2, dtype=torch.float64, device=x1.device)
    check(lib().ps_conv2d_stats_f32(ptr(x1), c1, ptr(x2), c2, ptr(wt), ptr(bias), ptr(y), n, m, f_in,
                                    t if t_in is None else t_in, t, ld, kf, kt, stride_f, dil_f, dil_t, pad_f, pad_t, f_out,
                                    int(transposed), ptr(stats), stream_ptr(x1.device)), "ps_conv2d_stats_f32")
    return y, stats


def activation_(x: torch.Tensor, kind: str, slope: Optional[torch.Tensor], t: int) -> torch.Tensor:
    """in place on [..., ld] rows."""
    require_device(x, "activation_")
    ld = x.shape[-1]
    check(lib().ps_activation_f32(ptr(x), ACT_KINDS[kind], ptr(slope), x.numel() // ld, t, ld, stream_ptr(x.device)),
          "ps_activation_f32")
    return x


def row_stats(x: torch.Tensor, t: int) -> torch.Tensor:
    """padded rows [N, rows, ld] -> [N, parts, 2] fp64 partial (sum, sum of squares) over the t valid frames."""
    require_device(x, "row_stats")
    n, rows, ld = x.shape
    out = torch.empty(n, lib().ps_row_stats_parts(), 2, dtype=torch.float64, device=x.device)
    check(lib().ps_row_stats_f64(ptr(x), ptr(out), n, rows, t, ld, stream_ptr(x.device)), "ps_row_stats_f64")
    return out


def norm_activation_(x4: torch.Tensor, t: int, pro: Prologue, corr_sum: float, corr_sq: float, kind: str,
                     slope: Optional[torch.Tensor]) -> torch.Tensor:
    """gLN over [CH, F, T] + activation, in place on [N, CH, F, ld]."""
    require_device(x4, "norm_activation_")
    n, ch, f, ld = x4.shape
    check(lib().ps_norm_activation_f32(ptr(x4), C.byref(pro), float(corr_sum), float(corr_sq), f, ACT_KINDS[kind],
                                       ptr(slope), n, ch * f, t, ld, stream_ptr(x4.device)), "ps_norm_activation_f32")
    return x4


def real_mask(feats: torch.Tensor, mask: torch.Tensor, mask_act: str = "linear") -> torch.Tensor:
    """feats * act(mask) on padded rows."""
    require_device(feats, "real_mask")
    if mask.shape != feats.shape:
        raise RuntimeError("real_mask: feats and mask must have one shape")
    out = torch.empty_like(feats)
    ldt = feats.shape[-1]
    check(lib().ps_real_mask_f32(ptr(feats), ptr(mask), ptr(out), feats.numel() // ldt, ldt, _abi.PS_ACT[mask_act],
                                 stream_ptr(feats.device)), "ps_real_mask_f32")
    return out


def fill_span(x: torch.Tensor, axis: int, lo: int, hi: int, value: float) -> torch.Tensor:
    """[N, rows, ld] -> a copy with rows [lo, hi) (axis 1) or frames [lo, hi) (axis 2) set to `value` (SpecAugment)."""
    require_device(x, "fill_span")
    n, rows, ld = x.shape
    y = torch.empty_like(x)
    check(lib().ps_fill_span_f32(ptr(x), ptr(y), n, rows, ld, axis, int(lo), int(hi), float(value), stream_ptr(x.device)),
          "ps_fill_span_f32")
    return y


def magnitude(x: torch.Tensor, t: int, drop_first: bool, log1p: bool, kind: Optional[str] = None) -> torch.Tensor:
    """[re rows; im rows] padded [N,2H,ldt] -> |.| (or log1p|.|, or the power with kind="power"/"power_eps")
    padded [N,H-drop,ldt]."""
    require_device(x, "magnitude")
    n, c2, ldt = x.shape
    half = c2 // 2
    y = torch.zeros(n, half - int(drop_first), ldt, dtype=torch.float32, device=x.device)
    k = {"power": 2, "power_eps": 3}[kind] if kind else int(log1p)
    check(lib().ps_magnitude_f32(ptr(x), ptr(y), n, half, int(drop_first), k, t, ldt, stream_ptr(x.device)),
          "ps_magnitude_f32")
    return y


def add_(dst: torch.Tensor, other: torch.Tensor) -> torch.Tensor:
    """dst += other (same shape, contiguous)."""
    require_device(dst, "add_")
    if dst.shape != other.shape or not dst.is_contiguous() or not other.is_contiguous():
        raise RuntimeError("add_: contiguous tensors of one shape")
    check(lib().ps_add_f32(ptr(dst), ptr(other), ptr(dst), dst.numel(), stream_ptr(dst.device)), "ps_add_f32")
    return dst


def self_attention(qkv: torch.Tensor, e: int, heads: int, q: int, q_stride: int, length: int, pos_stride: int,
                   causal: bool = False) -> torch.Tensor:
    """qkv padded [N,3E,ld] -> attention output [N,E,ld] (sequence (n,q): positions at q*q_stride + p*pos_stride)."""
    require_device(qkv, "self_attention")
    n, rows, ld = qkv.shape
    if rows != 3 * e:
        raise RuntimeError("self_attention: qkv must be [N, 3E, ld]")
    out = torch.empty(n, e, ld, dtype=torch.float32, device=qkv.device)
    check(lib().ps_self_attention_f32(ptr(qkv), ptr(out), n, e, heads, q, q_stride, length, pos_stride, ld, int(causal),
                                      stream_ptr(qkv.device)), "ps_self_attention_f32")
    return out


def add_position(x: torch.Tensor, pe: torch.Tensor, q: int, q_stride: int, length: int, pos_stride: int) -> torch.Tensor:
    """x padded [N,E,ld] + pe[pos][c] at every sequence position."""
    require_device(x, "add_position")
    n, e, ld = x.shape
    y = x.clone()
    check(lib().ps_add_position_f32(ptr(x), ptr(pe), ptr(y), n, e, q, q_stride, length, pos_stride, ld,
                                    stream_ptr(x.device)), "ps_add_position_f32")
    return y


def wave_moments(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    """Rows of two waveform batches [R, L] (last-dim contiguous) -> fp64 moments [R, 5] = (sum a, sum b, sum a^2,
    sum b^2, sum ab) in one streaming pass (ps_wave_moments_f64)."""
    require_device(a, "wave_moments")
    require_device(b, "wave_moments")
    if a.dim() != 2 or a.shape != b.shape or a.dtype != torch.float32 or b.dtype != torch.float32:
        raise RuntimeError("wave_moments: two fp32 tensors of the same shape [R, L]")
    if a.stride(1) != 1:
        a = a.contiguous()
    if b.stride(1) != 1:
        b = b.contiguous()
    r, length = a.shape
    chunks = lib().ps_wave_moments_chunks(length)
    part = torch.empty(r, chunks, 5, dtype=torch.float64, device=a.device)
    check(lib().ps_wave_moments_f64(ptr(a), ptr(b), ptr(part), r, length, a.stride(0), b.stride(0),
                                    stream_ptr(a.device)), "ps_wave_moments_f64")
    return part.sum(1)


def lstm_cell(gates: torch.Tensor, c: torch.Tensor, h: torch.Tensor, hidden: int, dirs: int, t: int) -> None:
    """One cell update per (unit, frame): gates padded [N,D*4H,ld] (complete pre-activations), c in place, h out
    (both [N,D*H,ld'] rows, possibly views into larger row blocks)."""
    require_device(gates, "lstm_cell")
    n = gates.shape[0]
    if gates.shape[1] != dirs * 4 * hidden or tuple(c.shape[:2]) != (n, dirs * hidden) or c.shape != h.shape \
            or c.stride(1) != h.stride(1):
        raise RuntimeError("lstm_cell: gates [N,D*4H,ld], c/h [N,D*H,ld']")
    if n > 1 and (not c.is_contiguous() or not h.is_contiguous()):
        raise RuntimeError("lstm_cell: state views are supported for N = 1 only")
    check(lib().ps_lstm_cell_f32(ptr(gates), ptr(c), ptr(h), n, hidden, dirs, t, gates.shape[2], c.stride(1),
                                 stream_ptr(gates.device)), "ps_lstm_cell_f32")


def film_apply(x: torch.Tensor, scale_bias: torch.Tensor, t: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """scale_bias padded [N,2C,ldt] (scale rows, then bias rows), x padded [N,C,ldt] -> scale * x + bias."""
    require_device(x, "film_apply")
    n, c, ldt = x.shape
    if tuple(scale_bias.shape) != (n, 2 * c, ldt):
        raise RuntimeError("film_apply: scale_bias must be [N, 2C, ldt]")
    y = out if out is not None else torch.empty_like(x)
    check(lib().ps_film_apply_f32(ptr(x), ptr(scale_bias), ptr(y), n, c, t, ldt, stream_ptr(x.device)),
          "ps_film_apply_f32")
    return y


def embed_bias(dvec: torch.Tensor, w_embed: torch.Tensor, normalize: bool) -> torch.Tensor:
    require_device(dvec, "embed_bias")
    require_weight(w_embed, dvec, "embed_bias")
    n, e = dvec.shape
    m = w_embed.shape[0]
    out = torch.empty(n, m, dtype=torch.float32, device=dvec.device)
    check(lib().ps_embed_bias_f32(ptr(dvec.contiguous()), ptr(w_embed), ptr(out), n, e, m, int(normalize),
                                  stream_ptr(dvec.device)), "ps_embed_bias_f32")
    return out


_EYE = {}


def l2_normalize(dvec: torch.Tensor) -> torch.Tensor:
    """F.normalize(dvec, p=2, dim=1) through ps_embed_bias_f32 with an identity weight (exact: every output is one
    product with 1 plus zeros)."""
    require_device(dvec, "l2_normalize")
    e = dvec.shape[1]
    key = (str(dvec.device), e)
    if key not in _EYE:
        _EYE[key] = torch.eye(e, dtype=torch.float32, device=dvec.device)
    return embed_bias(dvec.float(), _EYE[key], True)


def conv_tasnet(blocks: "C.Array[TcnBlock]", n_blocks: int, x_pad: torch.Tensor, t: int, c: int, h: int,
                dvec: Optional[torch.Tensor], embed_norm: bool,
                workspace: Optional[torch.Tensor] = None, x_amax: Optional[torch.Tensor] = None,
                bf16_rows: bool = False) -> torch.Tensor:
    """Run the whole masker on padded input [N,C,ldt]; returns padded mask logits [N,C,ldt] (the frames beyond T are
    not written).  x_amax [N, parts]: per utterance, values whose maximum bounds |x_pad[n]| -- the range blocks in the
    fp16x2 arithmetic scale their input by (without it they measure it with one pass over x_pad)."""
    require_device(x_pad, "conv_tasnet", allow_bf16=True)
    n, _, ldt = x_pad.shape
    if bf16_rows and x_pad.dtype == torch.float32:
        # fp32 rows in, fp32 rows out, the residual stream in between as bf16 rows (two dtype casts around the stack)
        return conv_tasnet(blocks, n_blocks, x_pad.to(torch.bfloat16), t, c, h, dvec, embed_norm, workspace).float()
    if x_pad.dtype == torch.bfloat16:
        # the residual stream as bf16 rows (BASELINE config 3's arithmetic): every block in the bf16 arithmetic
        need = lib().ps_conv_tasnet_workspace_bytes(n, c, h, t)
        if workspace is None or workspace.numel() < need:
            workspace = torch.zeros(need, dtype=torch.uint8, device=x_pad.device)
        out = torch.empty_like(x_pad)
        check(lib().ps_conv_tasnet_bf16_rows(blocks, n_blocks, ptr(x_pad), ptr(out), ptr(dvec), int(embed_norm), n, t, ldt,
                                             ptr(workspace), workspace.numel(), stream_ptr(x_pad.device)),
              "ps_conv_tasnet_bf16_rows")
        return out
    need = lib().ps_conv_tasnet_workspace_bytes(n, c, h, t)
    if workspace is None or workspace.numel() < need:
        workspace = torch.zeros(need, dtype=torch.uint8, device=x_pad.device)
    out = torch.empty_like(x_pad)
    parts = 0
    if x_amax is not None:
        require_device(x_amax, "conv_tasnet (x_amax)")
        if x_amax.dim() != 2 or x_amax.shape[0] != n or not x_amax.is_contiguous():
            raise ValueError(f"conv_tasnet: x_amax must be a contiguous [N={n}, parts] tensor, got {tuple(x_amax.shape)}")
        parts = x_amax.shape[1]
    check(lib().ps_conv_tasnet_ranged_f32(blocks, n_blocks, ptr(x_pad), ptr(out), ptr(dvec), int(embed_norm), n, t, ldt,
                                          ptr(workspace), workspace.numel(), ptr(x_amax), parts,
                                          stream_ptr(x_pad.device)), "ps_conv_tasnet_ranged_f32")
    return out
