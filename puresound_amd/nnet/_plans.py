"""Kernel-side parameter layouts ("plans") shared by the recurrent maskers.

A plan holds the packed / transposed fp32 copies of a module's parameters that the HIP kernels consume, keyed by a
fingerprint of the parameters so that load_state_dict / .to(device) / in-place edits invalidate it.  Plans hold
raw device pointers through the tensors they keep alive; they are never pickled.
"""
from typing import Optional

import torch
import torch.nn as nn

from .. import hip


import os

FUSE_PROJ_LN = os.environ.get("PS_FUSE_PROJ_LN", "1") == "1"   # 0: separate projection GEMM + LayerNorm kernels
FUSE_PROJ_LN_MAX_FRAMES = int(os.environ.get("PS_FUSE_PROJ_LN_MAX_FRAMES", "8192"))
FUSE_GEMM_LN = os.environ.get("PS_FUSE_GEMM_LN", "1") == "1"    # 0: projection GEMM and LayerNorm + skip as two launches
FMAJOR_LSTM = os.environ.get("PS_FMAJOR_LSTM", "1") == "1"      # 0: H = 128 recurrences stay on the channel-major kernels


# Arithmetic of the LSTM input projections (the large GEMMs of the recurrent maskers: [4H*D x C] over every frame):
# "fp32" = v_mfma_f32 on fp32 operands, "bf16x3" = fp32-accurate 3 x bf16 split, "bf16" = operands rounded to bf16 (what
# BASELINE.json names for the DPRNN configuration).  A per-module attribute, like TCN.gemm_precision: every PlanCache
# module carries `gemm_precision` and hands it to the LSTM plans it builds (no process-wide switch).
# "fp16x2" (the TCN blocks' default: two fp16 terms, three products) needs the range of the GEMM's input per utterance:
# lstm_path takes it from the caller (`amax`: the partial maxima the projection + LayerNorm row kernel of the previous
# recurrence left behind) or measures it (hip.absmax, one pass over the rows: the first GEMM of a masker).
_PLANES = {"fp32": 0, "bf16": 1, "bf16x3": 3, "fp16x2": 2}


def param_signature(module: nn.Module, device) -> tuple:
    sig = [(t.data_ptr(), t._version) for t in list(module.parameters()) + list(module.buffers())]
    sig.append(module.training)
    sig.append(str(device))
    return tuple(sig)


class PlanCache:
    """Mixin: `self._plan_get(device, builder)` caches builder(device) until a parameter changes."""

    _plan = None
    _plan_sig = None
    # arithmetic of the GEMMs (and, where kernels for it exist, the recurrent products) built by this module (see _PLANES).
    # Round 4: "fp16x2" is the default here as it has been for the TCN blocks since round 2 -- fp32-class results (two fp16
    # terms per operand, three products, fp32 accumulation: 1e-6 against exact fp32 products on every preset measured) at
    # 1.6-3.4 x the speed on the egs presets; set_gemm_precision("fp32") selects exact fp32 products.
    gemm_precision = "fp16x2"

    def set_gemm_precision(self, name: str):
        """"fp32" | "bf16x3" | "fp16x2" | "bf16" for the LSTM input projections of this module and of every module below
        it."""
        if name not in _PLANES:
            raise ValueError(f"gemm precision must be one of {sorted(_PLANES)}")
        for m in self.modules():
            if isinstance(m, PlanCache):
                m.gemm_precision = name
                m._plan = None
        return self

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_plan"] = None
        state["_plan_sig"] = None
        if "_last_amax" in state:   # (DPRNN: a device tensor handed from the block stack to the output conv within one call)
            state["_last_amax"] = None
        return state

    def _plan_get(self, device, builder):
        sig = param_signature(self, device) + (self.gemm_precision,)
        if self._plan is None or self._plan_sig != sig:
            self._plan = builder(device)
            self._plan_sig = sig
        return self._plan


def _f32(t: torch.Tensor, device) -> torch.Tensor:
    return t.detach().to(dtype=torch.float32, device=device).contiguous()


def lstm_plan(lstm: nn.LSTM, device, gemm: str = "fp32") -> dict:
    """nn.LSTM(num_layers=1, batch_first=True): input projection rows stacked over directions (for one
    ps_conv1x1_f32), summed biases, W_hh transposed per direction."""
    if lstm.num_layers != 1 or not lstm.batch_first or lstm.proj_size != 0 or not lstm.bias:
        raise NotImplementedError("HIP LSTM: num_layers=1, batch_first=True, bias=True, no projection")
    dirs = 2 if lstm.bidirectional else 1
    hid = lstm.hidden_size
    wih, bias, whh = [], [], []
    for suf in ["", "_reverse"][:dirs]:
        wih.append(_f32(getattr(lstm, "weight_ih_l0" + suf), device))
        bias.append(_f32(getattr(lstm, "bias_ih_l0" + suf), device) + _f32(getattr(lstm, "bias_hh_l0" + suf), device))
        whh.append(_f32(getattr(lstm, "weight_hh_l0" + suf), device).t().contiguous())
    return dict(wih=hip.pack_wt(torch.cat(wih, 0)), wih_rows=torch.cat(wih, 0).contiguous(), wih_planes={},
                bias=torch.cat(bias).contiguous(),
                whh_t=torch.stack(whh).contiguous(), H=hid, D=dirs, rows=dirs * 4 * hid, I=lstm.input_size,
                planes=_PLANES[gemm])


def rnn_plan(rnn: nn.Module, device) -> dict:
    """nn.GRU / nn.RNN(tanh; num_layers=1, batch_first=True) for ps_rnn_f32: input projection rows stacked over directions,
    b_hh folded into the projection's bias except the GRU's n gate (its hidden bias sits inside the reset product: bhn)."""
    kind = "GRU" if isinstance(rnn, nn.GRU) else "RNN"
    if kind == "RNN" and getattr(rnn, "nonlinearity", "tanh") != "tanh":
        raise NotImplementedError("nn.RNN on HIP: tanh cells (the reference's SingleRNN builds no other)")
    if rnn.num_layers != 1 or not rnn.batch_first:
        raise NotImplementedError("HIP recurrences: one layer, batch_first")
    hid, ng = rnn.hidden_size, (3 if kind == "GRU" else 1)
    wih, bias, whh, bhn = [], [], [], []
    for suf in ([""] if not rnn.bidirectional else ["", "_reverse"]):
        b_hh = _f32(getattr(rnn, "bias_hh_l0" + suf), device)
        fold = b_hh.clone()
        if kind == "GRU":
            fold[2 * hid:] = 0.0
            bhn.append(b_hh[2 * hid:].clone())
        wih.append(_f32(getattr(rnn, "weight_ih_l0" + suf), device))
        bias.append(_f32(getattr(rnn, "bias_ih_l0" + suf), device) + fold)
        whh.append(_f32(getattr(rnn, "weight_hh_l0" + suf), device).t().contiguous())
    rows = torch.cat(wih, 0).contiguous()
    return dict(kind=kind, wih=hip.pack_wt(rows), rows=rows.shape[0], bias=torch.cat(bias).contiguous(),
                whh_t=torch.stack(whh).contiguous(), bhn=torch.stack(bhn).contiguous() if bhn else None, H=hid,
                D=2 if rnn.bidirectional else 1, I=rnn.input_size)


def linear_plan(lin: nn.Module, device) -> dict:
    """nn.Linear or nn.Conv1d(k=1)."""
    w = lin.weight
    if w.dim() == 3:
        w = w[:, :, 0]
    return dict(wt=hip.pack_wt(_f32(w, device)), bias=None if lin.bias is None else _f32(lin.bias, device),
                M=w.shape[0], K=w.shape[1], w_rows=_f32(w, device))


def layernorm_plan(ln: nn.Module, device) -> dict:
    """nn.LayerNorm(C) (weight/bias, eps) or ChanLN (gamma/beta, eps 1e-8)."""
    if isinstance(ln, nn.LayerNorm):
        if len(ln.normalized_shape) != 1 or ln.weight is None:
            raise NotImplementedError("HIP LayerNorm: one normalised dim with affine parameters")
        return dict(gamma=_f32(ln.weight, device), beta=_f32(ln.bias, device), eps=float(ln.eps))
    return dict(gamma=_f32(ln.gamma, device), beta=_f32(ln.beta, device), eps=float(ln.eps))


def _proj_norm(x, hseq, t, rnn, proj, norm, amax, skip=True):
    """x + LN(proj(h)) behind a recurrence (second half of lstm_path); skip=False: LN(proj(h)) alone."""
    res = x if skip else None
    n, _, ldt = x.shape
    dev = x.device
    # the fused projection + LayerNorm kernel is the short-row kernel (16-frame workgroups): on long rows (the 2-D
    # maps of DPCRN / DPARN: F * ld frames) the MFMA GEMM plus a LayerNorm pass is faster
    if proj["M"] <= 256 and FUSE_PROJ_LN and t <= FUSE_PROJ_LN_MAX_FRAMES:
        if amax is not None and rnn["planes"] == 2 and t >= 128:
            try:   # the row kernel hands the next GEMM its input range for free
                y, _, amax[0] = hip.proj_layernorm(hseq, t, proj["wt"], proj["bias"], proj["M"], norm["gamma"], norm["beta"],
                                                   norm["eps"], res, want_amax=True)
                return y
            except RuntimeError:
                amax[0] = None
        y, _ = hip.proj_layernorm(hseq, t, proj["wt"], proj["bias"], proj["M"], norm["gamma"], norm["beta"], norm["eps"], res)
        return y
    if (FUSE_GEMM_LN and rnn["planes"] == 2 and proj["M"] == 128 and proj["K"] % 32 == 0
            and hip.conv1x1_f16x2_ln_ok(n, proj["K"], 128, t)):
        # projection, LayerNorm and skip as ONE launch: the fp16x2 GEMM with the norm as its epilogue (|h| < 1 is its range)
        if "f16x2_ln" not in proj:
            w256 = torch.zeros(256, proj["K"], dtype=torch.float32, device=dev)
            w256[:128] = proj["w_rows"]
            proj["f16x2_ln"] = hip.pack_wt_f16x2(w256)
        wf, we = proj["f16x2_ln"]
        if amax is not None:   # the maxima of |y| ride along: the next recurrence's GEMM needs no pass of its own over y
            y, amax[0] = hip.conv1x1_f16x2_ln(hseq, t, wf, we, 128, proj["bias"], norm["gamma"], norm["beta"], norm["eps"], res,
                                              x_bound=1.0, want_amax=True)
            return y
        return hip.conv1x1_f16x2_ln(hseq, t, wf, we, 128, proj["bias"], norm["gamma"], norm["beta"], norm["eps"], res, x_bound=1.0)
    p = torch.empty(n, proj["M"], ldt, dtype=torch.float32, device=dev)
    if rnn["planes"] == 2 and proj["K"] >= 64:
        # the recurrence's arithmetic for its projection too: h is an LSTM output, |h| < 1 is its range
        if "f16x2" not in proj:
            proj["f16x2"] = hip.pack_wt_f16x2(proj["w_rows"])
        hip.conv1x1_f16x2(hseq, t, proj["f16x2"][0], proj["f16x2"][1], proj["M"], None, proj["bias"], out=p, x_bound=1.0)
    else:
        hip.conv1x1(hseq, t, proj["wt"], proj["M"], None, proj["bias"], out=p)
    return hip.chan_layernorm(p, t, norm["gamma"], norm["beta"], norm["eps"], res=res)


def _h_rows(rnn: dict, x: torch.Tensor, t: int, q: int, steps: int):
    """Output rows of a recurrence.  When the sequences do not cover the span of t frames (the 2-D maps of DPCRN / DPARN:
    F rows of ld frames, pad frames between them) the recurrence never writes the pad frames, and what follows it --
    projection, LayerNorm, the next GEMM's range maximum -- runs over the whole span: the rows then come from a zeroed
    buffer kept with the plan (pads stay zero from call to call), so stale memory (a NaN, an Inf) cannot reach the maximum.
    None = let the kernel wrapper allocate (every frame is written)."""
    if q * steps >= t:
        return None
    n, _, ldt = x.shape
    key = (n, ldt, x.device)
    bufs = rnn.setdefault("h_rows", {})
    if key not in bufs:
        bufs.clear()
        bufs[key] = torch.zeros(n, rnn["D"] * rnn["H"], ldt, dtype=torch.float32, device=x.device)
    return bufs[key]


def lstm_path(x: torch.Tensor, t: int, rnn: dict, proj: dict, norm: dict, q: int, q_stride: int, steps: int,
              step_stride: int, h0=None, c0=None, want_state: bool = False, state_shift: int = 0, state_out=None,
              amax: Optional[list] = None, skip: bool = True):
    """x + LN(proj(LSTM(x))) on padded [N,C,ldt] (dprnn.py:154-172, skim.py:215-227) -> (x', final states).
    `amax`: a one-element list carried from recurrence to recurrence in the fp16x2 arithmetic -- on entry the partial maxima
    of |x| per utterance (or None: measured here), on return those of x' (None when the kernel that made x' has none)."""
    n, _, ldt = x.shape
    dev = x.device
    planes = rnn["planes"]
    x_amax = amax[0] if amax is not None else None
    if amax is not None:
        amax[0] = None
    if planes == 2 and rnn["I"] >= 64:
        if "wih_f16x2" not in rnn:
            rnn["wih_f16x2"] = hip.pack_wt_f16x2(rnn["wih_rows"])
        wf, we = rnn["wih_f16x2"]
        # H = 128 without carried states (the bottleneck LSTMs of DPCRN / DPARN): gate pre-activations frame-major and the
        # 16-sequence fp16x2 recurrence that streams them (ps_lstm_fmajor_f16x2_f32)
        if (FMAJOR_LSTM and rnn["H"] == 128 and h0 is None and c0 is None and not want_state and state_out is None
                and state_shift == 0 and hip.lstm_fmajor_ok(n, ldt, rnn["H"], rnn["D"], q, q_stride, steps, step_stride)
                and hip.conv1x1_f16x2_fmajor_ok(n, rnn["I"], rnn["rows"], t, ldt)):
            gx_fm = hip.conv1x1_f16x2_fmajor(x, t, wf, we, rnn["rows"], rnn["bias"],
                                             x_amax=x_amax if x_amax is not None else hip.absmax(x, t))
            hseq = hip.lstm_fmajor(gx_fm, rnn["whh_t"], rnn["H"], rnn["D"], q, q_stride, steps, step_stride,
                                   out=_h_rows(rnn, x, t, q, steps))
            return _proj_norm(x, hseq, t, rnn, proj, norm, amax, skip), None
        # H = 256 (SkiM's segment LSTMs): W_hh streamed from L2 in fragment order, states carried (ps_lstm_fmajor_h256_f16x2_f32)
        if (FMAJOR_LSTM and rnn["H"] in (256, 192) and hip.lstm_fmajor_h256_ok(n, ldt, rnn["D"], q, q_stride, steps, step_stride)
                and hip.conv1x1_f16x2_fmajor_ok(n, rnn["I"], rnn["rows"], t, ldt)):
            if "whh_h256" not in rnn:
                rnn["whh_h256"] = hip.pack_whh_h256(rnn["whh_t"])
            gx_fm = hip.conv1x1_f16x2_fmajor(x, t, wf, we, rnn["rows"], rnn["bias"],
                                             x_amax=x_amax if x_amax is not None else hip.absmax(x, t))
            hseq, state = hip.lstm_fmajor_h256(gx_fm, rnn["whh_h256"][0], rnn["whh_h256"][1], rnn["D"], q, q_stride, steps,
                                               step_stride, h0, c0, want_state, state_shift, state_out,
                                               out=_h_rows(rnn, x, t, q, steps))
            return _proj_norm(x, hseq, t, rnn, proj, norm, amax, skip), state
        gx = torch.empty(n, rnn["rows"], ldt, dtype=torch.float32, device=dev)
        hip.conv1x1_f16x2(x, t, wf, we, rnn["rows"], None, rnn["bias"], out=gx,
                          x_amax=x_amax if x_amax is not None else hip.absmax(x, t))
    elif planes and rnn["I"] >= 64:
        gx = torch.empty(n, rnn["rows"], ldt, dtype=torch.float32, device=dev)
        planes = 3 if planes == 2 else planes
        if planes not in rnn["wih_planes"]:
            rnn["wih_planes"][planes] = hip.pack_wt_bf16(rnn["wih_rows"], planes)
        hip.conv1x1_bf16(x, t, rnn["wih_planes"][planes], rnn["rows"], None, rnn["bias"], out=gx)
    else:
        gx = torch.empty(n, rnn["rows"], ldt, dtype=torch.float32, device=dev)
        hip.conv1x1(x, t, rnn["wih"], rnn["rows"], None, rnn["bias"], out=gx)
    hseq, state = hip.lstm(gx, rnn["whh_t"], rnn["H"], rnn["D"], q, q_stride, steps, step_stride, h0, c0, want_state,
                           state_shift, state_out, f16x2=rnn["planes"] == 2, out=_h_rows(rnn, x, t, q, steps))
    return _proj_norm(x, hseq, t, rnn, proj, norm, amax, skip), state
