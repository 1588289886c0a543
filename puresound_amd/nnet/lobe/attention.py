"""Attention lobes (mirror of puresound/nnet/lobe/attention.py:8-232): parameter trees under the reference's keys and
the HIP driver of the post-norm transformer layer used by DPARN (improved=False):

    [x + pe] -> in_proj GEMM -> ps_self_attention_f32 -> out_proj + residual + LayerNorm (one kernel)
             -> Linear + ReLU + Linear + residual (two GEMMs, the ReLU is the second one's prologue) -> LayerNorm
"""
import math
from typing import Optional

import torch
import torch.nn as nn

from ... import hip
from .._plans import PlanCache, _f32, layernorm_plan, linear_plan, lstm_plan


class PositionalEncoding(nn.Module):
    """lobe/attention.py:8-35; `pe` [max_len, 1, d_model] is a persistent buffer (it is in the checkpoints)."""

    def __init__(self, d_model: int, dropout: float = 0.1, max_len: int = 5000):
        super().__init__()
        if d_model % 2 != 0:
            raise ValueError(f"Cannot use sin/cos positional encoding with odd dim (got dim={d_model})")
        self.dropout = nn.Dropout(p=dropout)
        position = torch.arange(max_len).unsqueeze(1)
        div_term = torch.exp(torch.arange(0, d_model, 2) * (-math.log(10000.0) / d_model))
        pe = torch.zeros(max_len, 1, d_model)
        pe[:, 0, 0::2] = torch.sin(position * div_term)
        pe[:, 0, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("PositionalEncoding is applied by ps_add_position_f32 inside MhaSelfAttenLayer")


class MHA(nn.Module):
    """lobe/attention.py:38-112: holds nn.MultiheadAttention(bias=False, batch_first=True)."""

    def __init__(self, embed_dim: int, heads: int = 1):
        super().__init__()
        self.atten = nn.MultiheadAttention(embed_dim=embed_dim, num_heads=heads, dropout=0, batch_first=True,
                                           bias=False)

    def forward(self, *args, **kwargs):
        raise NotImplementedError("MHA runs inside MhaSelfAttenLayer on the HIP path")


class MhaSelfAttenLayer(PlanCache, nn.Module):
    """Transformer encoder block (lobe/attention.py:115-232)."""

    def __init__(self, feats_dim: int, hidden_dim: int, nhead: int, dropout: float = 0.0, improved: bool = False,
                 bidirectional: bool = False, position_encoding: bool = True):
        super().__init__()
        self.improved = improved
        self.bidirectional = bidirectional
        self.position_encoding = position_encoding
        self.feats_dim, self.nhead = feats_dim, nhead
        self.self_atten = MHA(feats_dim, heads=nhead)
        self.self_atten_dropout = nn.Dropout(p=dropout)
        self.norm1 = nn.LayerNorm(feats_dim)
        if not improved:
            if self.bidirectional:
                print("Ignored bidirectional option since no LSTM here.")
            if position_encoding:
                self.pos = PositionalEncoding(d_model=feats_dim, dropout=dropout)
            self.feedforward = nn.Sequential(nn.Linear(feats_dim, hidden_dim), nn.ReLU(), nn.Dropout(p=dropout),
                                             nn.Linear(hidden_dim, feats_dim), nn.Dropout(p=dropout))
        else:
            if position_encoding:
                print("Ignored position_encoding option here replaced by LSTM modeling.")
            self.recurrent = nn.LSTM(feats_dim, hidden_dim, bidirectional=bidirectional, batch_first=True)
            if bidirectional:
                hidden_dim *= 2
            self.feedforward = nn.Sequential(nn.ReLU(), nn.Dropout(p=dropout), nn.Linear(hidden_dim, feats_dim),
                                             nn.Dropout(p=dropout))
        self.norm2 = nn.LayerNorm(feats_dim)

    def _build(self, device):
        drops = [m for m in [self.self_atten_dropout] + list(self.feedforward) if isinstance(m, nn.Dropout)]
        if self.training and any(d.p > 0 for d in drops):
            raise RuntimeError("MhaSelfAttenLayer: dropout is active; the HIP path is inference only -- call .eval()")
        at = self.self_atten.atten
        p = dict(w_in=hip.pack_wt(_f32(at.in_proj_weight, device)),
                 out=dict(wt=hip.pack_wt(_f32(at.out_proj.weight, device)), M=self.feats_dim),
                 norm1=layernorm_plan(self.norm1, device), norm2=layernorm_plan(self.norm2, device))
        if self.improved:
            # the "improved" transformer (lobe/attention.py:170-183): the first feed-forward Linear is an LSTM over the
            # sequence -- input projection GEMM + ps_lstm_f32 -- then ReLU -> Linear as the second GEMM's prologue
            p["rnn"] = lstm_plan(self.recurrent, device)
            p["ff2"] = linear_plan(self.feedforward[2], device)
        else:
            p["ff1"] = linear_plan(self.feedforward[0], device)
            p["ff2"] = linear_plan(self.feedforward[3], device)
        if self.position_encoding and not self.improved:
            p["pe"] = _f32(self.pos.pe[:, 0, :], device)            # [max_len, E]
        if self.gemm_precision == "fp16x2" and not self.improved and self.feats_dim == 128:
            # the layer's five GEMMs in the fp16x2 arithmetic (two fp16 terms per operand, fp32 accumulation), the two
            # LayerNorms as epilogues of theirs.  Input ranges without extra passes: a LayerNorm's output is bounded by
            # sqrt(C) max|gamma| + max|beta| (host constants, read once here); q / k / v and the hidden layer leave their
            # maxima behind (y_amax); the attention output is a convex combination of v rows.
            def pad256(w):
                w256 = torch.zeros(256, w.shape[1], dtype=torch.float32, device=device)
                w256[:w.shape[0]] = w
                return hip.pack_wt_f16x2(w256)

            def ln_bound(ln):
                return float(math.sqrt(self.feats_dim) * ln["gamma"].abs().max() + ln["beta"].abs().max()) * 1.0001

            w_ff1 = _f32(self.feedforward[0].weight, device)
            p["f16x2"] = dict(w_in=hip.pack_wt_f16x2(_f32(at.in_proj_weight, device)),
                              out=pad256(_f32(at.out_proj.weight, device)),
                              ff1=hip.pack_wt_f16x2(w_ff1), ff2=pad256(_f32(self.feedforward[3].weight, device)),
                              zero_slope=torch.zeros(1, dtype=torch.float32, device=device),
                              bound1=ln_bound(p["norm1"]), bound2=ln_bound(p["norm2"]))
        return p

    def _forward_f16x2(self, x, src, frames, q, q_stride, length, pos_stride, causal, p, f, x_amax, x_bound):
        """The layer in the fp16x2 arithmetic: in_proj GEMM -> attention -> (out_proj + residual + norm1) as one launch ->
        Linear -> (ReLU + Linear + residual + norm2) as one launch."""
        n, e, ld = x.shape
        hid = p["ff1"]["M"]
        if self.position_encoding:   # |pe| <= 1 on top of whatever range the caller knew
            x_bound = x_bound + 1.0 if x_bound > 0 else 0.0
            x_amax = x_amax + 1.0 if x_amax is not None else None
        if x_bound <= 0 and x_amax is None:
            x_amax = hip.absmax(x, frames)
        qkv, _, qkv_amax = hip.conv1x1_f16x2(x, frames, f["w_in"][0], f["w_in"][1], 3 * e, x_bound=x_bound,
                                             x_amax=None if x_bound > 0 else x_amax, want_amax=True,
                                             out=torch.empty(n, 3 * e, ld, dtype=torch.float32, device=x.device))
        att = hip.self_attention(qkv, e, self.nhead, q, q_stride, length, pos_stride, causal)
        n1, n2 = p["norm1"], p["norm2"]
        y = hip.conv1x1_f16x2_ln(att, frames, f["out"][0], f["out"][1], e, None, n1["gamma"], n1["beta"], n1["eps"], src,
                                 x_amax=qkv_amax, res_inside=True)
        h, _, h_amax = hip.conv1x1_f16x2(y, frames, f["ff1"][0], f["ff1"][1], hid, None, p["ff1"]["bias"], x_bound=f["bound1"],
                                         want_amax=True, out=torch.empty(n, hid, ld, dtype=torch.float32, device=x.device))
        relu = hip.make_prologue(0, True, None, 0.0, 0.0, None, None, f["zero_slope"])
        return hip.conv1x1_f16x2_ln(h, frames, f["ff2"][0], f["ff2"][1], e, p["ff2"]["bias"], n2["gamma"], n2["beta"], n2["eps"],
                                    y, x_amax=h_amax, pro=relu, res_inside=True)

    def forward_padded(self, x: torch.Tensor, frames: int, q: int, q_stride: int, length: int, pos_stride: int,
                       causal: bool = False, x_amax: Optional[torch.Tensor] = None, want_bound: bool = False,
                       x_bound: float = 0.0):
        """x padded [N, E, ld]: sequences (n, q) with `length` positions at q*q_stride + p*pos_stride.  x_amax: partial maxima
        of |x| / x_bound: a host bound on |x|, when the caller has them (fp16x2 arithmetic).  want_bound: return (y, bound) with bound >= max |y| when the
        layer knows one (its last operation is a LayerNorm), else None."""
        y, bound = self._forward_padded(x, frames, q, q_stride, length, pos_stride, causal, x_amax, x_bound)
        return (y, bound) if want_bound else y

    def _forward_padded(self, x, frames, q, q_stride, length, pos_stride, causal, x_amax, x_bound):
        p = self._plan_get(x.device, self._build)
        n, e, ld = x.shape
        new = lambda rows: torch.empty(n, rows, ld, dtype=torch.float32, device=x.device)  # noqa: E731
        src = x
        if self.position_encoding and self.improved:
            # the reference's forward calls self.pos whenever position_encoding is set, and the improved layer never builds
            # it (lobe/attention.py:165-168, 199-200): the same AttributeError
            raise AttributeError("'MhaSelfAttenLayer' object has no attribute 'pos'")
        if self.position_encoding:
            if length > p["pe"].shape[0]:
                raise RuntimeError("sequence longer than the positional table")
            x = hip.add_position(x, p["pe"], q, q_stride, length, pos_stride)
        f = p.get("f16x2")
        if (f is not None and p["ff1"]["M"] % 32 == 0 and hip.conv1x1_f16x2_ln_ok(n, e, 128, frames)
                and hip.conv1x1_f16x2_ln_ok(n, p["ff1"]["M"], 128, frames)):
            return self._forward_f16x2(x, src, frames, q, q_stride, length, pos_stride, causal, p, f, x_amax, x_bound), f["bound2"]
        qkv, _ = hip.conv1x1(x, frames, p["w_in"], 3 * e, out=new(3 * e))
        att = hip.self_attention(qkv, e, self.nhead, q, q_stride, length, pos_stride, causal)
        n1 = p["norm1"]
        y, _ = hip.proj_layernorm(att, frames, p["out"]["wt"], None, e, n1["gamma"], n1["beta"], n1["eps"], src,
                                  res_inside=True)
        if self.improved:
            rnn = p["rnn"]
            gx, _ = hip.conv1x1(y, frames, rnn["wih"], rnn["rows"], None, rnn["bias"], out=new(rnn["rows"]))
            h, _ = hip.lstm(gx, rnn["whh_t"], rnn["H"], rnn["D"], q, q_stride, length, pos_stride, None, None, False, 0, None)
        else:
            h, _ = hip.conv1x1(y, frames, p["ff1"]["wt"], p["ff1"]["M"], None, p["ff1"]["bias"], out=new(p["ff1"]["M"]))
        n2 = p["norm2"]
        pro = hip.make_prologue(0, False, None, 0.0, 0.0, None, None, None, pre_relu=True)
        s, _ = hip.conv1x1(h, frames, p["ff2"]["wt"], e, pro, p["ff2"]["bias"], res=y, out=new(e))
        return hip.chan_layernorm(s, frames, n2["gamma"], n2["beta"], n2["eps"]), None

    def forward(self, x: torch.Tensor, causal: bool = False, context_range: Optional[int] = None,
                return_atten_weight: bool = False):
        """x [N, C, T] -> [N, C, T] (lobe/attention.py:180-232)."""
        hip.require_device(x, "MhaSelfAttenLayer.forward")
        if context_range is not None or return_atten_weight:
            raise NotImplementedError("MhaSelfAttenLayer on HIP: context_range / return_atten_weight")
        t = x.shape[-1]
        xp = hip.pad_rows(x)
        # every batch entry is one sequence over its t frames
        return hip.unpad_rows(self.forward_padded(xp, t, 1, 0, t, 1, causal), t)
