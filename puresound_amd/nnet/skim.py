"""Skipping-memory LSTM masker on the HIP path (mirror of puresound/nnet/skim.py:11-469).

Same layout as dprnn.py: padded channel-major rows holding T' = S*K frames.  SegLSTM = the intra-segment
addressing of ps_lstm_f32 with per-segment initial/final states kept in the "state layout" [N][D*H][ldS] (one
frame per segment); MemLSTM runs on that state tensor directly as a sequence of S frames per utterance, and its
causal one-segment shift (including the reference's leak from the last segment of utterance n-1 into the first
of utterance n, skim.py:102-109) is the state_shift of the next SegLSTM call.
"""
from typing import List, Optional, Tuple

import torch
import torch.nn as nn

from .. import hip
from ..ops import op_module
from ._plans import PlanCache, _f32, layernorm_plan, linear_plan, lstm_path, lstm_plan
from .lobe.trivial import FiLM, Gate


class MemLSTM(PlanCache, nn.Module):
    """skim.py:11-43 (parameters), :45-114 (forward)."""

    def __init__(self, hidden_size: int, causal: bool = True, dropout: float = 0.0):
        super().__init__()
        self.hidden_size = hidden_size
        self.causal = causal
        self.input_size = hidden_size if causal else 2 * hidden_size
        self.bi_direct = not causal
        d = int(self.bi_direct) + 1
        self.h_net = nn.LSTM(self.input_size, self.hidden_size, num_layers=1, bidirectional=self.bi_direct,
                             batch_first=True)
        self.h_dropout = nn.Dropout(p=dropout)
        self.h_proj = nn.Linear(self.hidden_size * d, self.input_size)
        self.h_norm = nn.LayerNorm(self.input_size)
        self.c_net = nn.LSTM(self.input_size, self.hidden_size, num_layers=1, bidirectional=self.bi_direct,
                             batch_first=True)
        self.c_dropout = nn.Dropout(p=dropout)
        self.c_proj = nn.Linear(self.hidden_size * d, self.input_size)
        self.c_norm = nn.LayerNorm(self.input_size)

    def _build(self, device):
        if self.training and (self.h_dropout.p > 0 or self.c_dropout.p > 0):
            raise RuntimeError("MemLSTM: dropout is active; the HIP path is inference only -- call .eval()")
        return dict(h=(lstm_plan(self.h_net, device, self.gemm_precision), linear_plan(self.h_proj, device),
                       layernorm_plan(self.h_norm, device)),
                    c=(lstm_plan(self.c_net, device, self.gemm_precision), linear_plan(self.c_proj, device),
                       layernorm_plan(self.c_norm, device)))

    def step_plans(self, device) -> dict:
        """Streaming update (one LSTM step per stream, states carried): for each of the two nets, [W_ih | W_hh] as ONE weight
        over K = [x; h] with rows reordered unit-major (4u + g) for ps_lstm_gates_cell_f32, plus its projection and norm
        plans.  Unidirectional nets only (the causal SkiM)."""
        p = self._plan_get(device, self._build)
        if "units" not in p:
            units = {}
            for key, net in (("h", self.h_net), ("c", self.c_net)):
                if net.bidirectional:
                    raise NotImplementedError("merged-gate streaming update: causal MemLSTM only")
                w = torch.cat([_f32(net.weight_ih_l0, device), _f32(net.weight_hh_l0, device)], dim=1)
                hid = net.hidden_size
                order = (torch.arange(4, device=device).reshape(1, 4) * hid + torch.arange(hid, device=device).reshape(hid, 1)
                         ).reshape(-1)
                units[key] = dict(w_units=hip.pack_wt(w[order].contiguous()), bias_units=p[key][0]["bias"][order].contiguous(),
                                  proj=p[key][1], norm=p[key][2], H=hid, I=net.input_size)
            p["units"] = units
        return p["units"]

    def forward_state(self, h: torch.Tensor, c: torch.Tensor, s: int, h_states=None, c_states=None,
                      want_states: bool = False, per_frame_sequences: bool = False):
        """h, c: state layout [N, D*H, ldS] with S frames.  Offline: every utterance is one sequence over its S
        segments.  per_frame_sequences (streaming): every frame is its own one-step sequence with carried LSTM
        states.  Returns (h', c', h_states', c_states'); the causal shift is left to the consumer."""
        p = self._plan_get(h.device, self._build)
        outs, states = [], []
        for key, v, st in (("h", h, h_states), ("c", c, c_states)):
            h0, c0 = st if st is not None else (None, None)
            if per_frame_sequences:
                y, new = lstm_path(v, s, *p[key], q=s, q_stride=1, steps=1, step_stride=0, h0=h0, c0=c0,
                                   want_state=True)
            else:
                y, new = lstm_path(v, s, *p[key], q=1, q_stride=0, steps=s, step_stride=1, h0=h0, c0=c0,
                                   want_state=want_states)
            outs.append(y)
            states.append(new)
        return outs[0], outs[1], states[0], states[1]

    def forward(self, h, c, h_states=None, c_states=None, return_all: bool = False, streaming: bool = False):
        """Reference signature (skim.py:45-114): h, c [N,S,D,H] -> [D,N*S,H] (+ LSTM states [D,N,H] pairs)."""
        hip.require_device(h, "MemLSTM.forward")
        n, s, d, hid = h.shape
        to_state = lambda v: hip.pad_rows(v.reshape(n, s, d * hid).transpose(1, 2))  # noqa: E731
        conv = lambda st: None if st is None else tuple(hip.pad_rows(x.permute(1, 0, 2).reshape(n, d * hid, 1))  # noqa: E731
                                                        for x in st)
        ho, co, hs, cs = self.forward_state(to_state(h), to_state(c), s, conv(h_states), conv(c_states), True)

        def back(v):  # [N, D*H, ldS] -> [D, N*S, H]
            v = v[..., :s].reshape(n, d, hid, s).permute(1, 0, 3, 2).reshape(d, n * s, hid)
            if self.causal and not streaming:
                shifted = torch.zeros_like(v)
                shifted[:, 1:] = v[:, :-1]
                v = shifted
            return v.contiguous()

        unstate = lambda st: tuple(x[..., 0].reshape(n, d, hid).permute(1, 0, 2).contiguous() for x in st)  # noqa: E731
        if return_all:
            return back(ho), back(co), unstate(hs), unstate(cs)
        return back(ho), back(co)


class SegLSTM(PlanCache, nn.Module):
    """skim.py:168-229."""

    def __init__(self, input_size: int, hidden_size: int, causal: bool = True, dropout: float = 0.0):
        super().__init__()
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.bi_direct = not causal
        self.causal = causal
        self.lstm = nn.LSTM(input_size, hidden_size, num_layers=1, bidirectional=self.bi_direct, batch_first=True)
        self.drop = nn.Dropout(p=dropout)
        self.proj = nn.Linear(hidden_size * (int(self.bi_direct) + 1), input_size)
        self.norm = nn.LayerNorm(input_size)

    def _build(self, device):
        if self.training and self.drop.p > 0:
            raise RuntimeError("SegLSTM: dropout is active; the HIP path is inference only -- call .eval()")
        return (lstm_plan(self.lstm, device, self.gemm_precision), linear_plan(self.proj, device), layernorm_plan(self.norm, device))

    def step_plan(self, device):
        """Streaming step: [W_ih | W_hh] as ONE weight whose K axis is [x; h], so the gates of a frame step are a
        single GEMM followed by ps_lstm_cell_f32 (unidirectional only)."""
        if self.bi_direct:
            raise NotImplementedError("merged-gate streaming step: causal SegLSTM only")
        p = self._plan_get(device, self._build)
        if "w_gates" not in p[0]:
            w = torch.cat([_f32(self.lstm.weight_ih_l0, device), _f32(self.lstm.weight_hh_l0, device)], dim=1)
            p[0]["w_gates"] = hip.pack_wt(w)
            # unit-major rows (4u + g) for the fused gates + cell kernel
            h = self.hidden_size
            order = (torch.arange(4, device=device).reshape(1, 4) * h + torch.arange(h, device=device).reshape(h, 1)
                     ).reshape(-1)
            p[0]["w_units"] = hip.pack_wt(w[order].contiguous())
            p[0]["bias_units"] = p[0]["bias"][order].contiguous()
        return p

    def forward_padded(self, x: torch.Tensor, tp: int, q: int, k: int, h0, c0, state_shift: int = 0):
        """x padded [N,C,ldt] with q sequences of k contiguous frames per utterance; states in state layout."""
        p = self._plan_get(x.device, self._build)
        return lstm_path(x, tp, *p, q=q, q_stride=k, steps=k, step_stride=1, h0=h0, c0=c0, want_state=True,
                         state_shift=state_shift)

    def forward(self, x: torch.Tensor, h: Optional[torch.Tensor], c: Optional[torch.Tensor]):
        """Reference signature (skim.py:198-229): x [B,K,C], h/c [D,B,H] -> (x', h, c)."""
        hip.require_device(x, "SegLSTM.forward")
        b, k, _ = x.shape
        d, hid = int(self.bi_direct) + 1, self.hidden_size
        # every batch entry is an utterance with a single segment
        to_state = lambda v: None if v is None else hip.pad_rows(v.permute(1, 0, 2).reshape(b, d * hid, 1))  # noqa: E731
        y, (hl, cl) = self.forward_padded(hip.pad_rows(x.transpose(1, 2)), k, 1, k, to_state(h), to_state(c))
        back = lambda v: v[..., 0].reshape(b, d, hid).permute(1, 0, 2).contiguous()  # noqa: E731
        return hip.unpad_rows(y, k).transpose(1, 2).contiguous(), back(hl), back(cl)


def _out_size(ctor, x, aux, params):
    return (x[0], ctor["output_size"], x[2])


@op_module("skim_fwd", _out_size)
class SkiM(PlanCache, nn.Module):
    """Skipping memory LSTM (skim.py:251-469); constructor order as the reference (skim.py:280-294)."""

    def __init__(self, input_size: int, hidden_size: int, output_size: int, n_blocks: int = 2, seg_size: int = 20,
                 seg_overlap: bool = False, causal: bool = True, embed_dim: int = 0, embed_norm: bool = False,
                 embed_fusion: Optional[str] = None, block_with_embed: Optional[List] = None, dropout: float = 0.0):
        super().__init__()
        self.seg_size = seg_size
        self.seg_overlap = seg_overlap
        self.hidden_size = hidden_size
        self.input_size = input_size
        self.n_blocks = n_blocks
        self.causal = causal
        self.embed_dim = embed_dim
        self.embed_norm = embed_norm
        self.block_with_embed = block_with_embed

        self.seg_lstm = nn.ModuleList()
        if embed_dim == 0:
            for _ in range(n_blocks):
                self.seg_lstm.append(SegLSTM(input_size, hidden_size, causal=causal, dropout=dropout))
        else:
            self.seg_input_fusion = nn.ModuleList()
            for i in range(n_blocks):
                self.seg_lstm.append(SegLSTM(input_size, hidden_size, causal=causal, dropout=dropout))
                if block_with_embed[i]:
                    if embed_fusion.lower() == "film":
                        self.seg_input_fusion.append(FiLM(input_size, embed_dim, input_norm=True))
                    elif embed_fusion.lower() == "gate":
                        self.seg_input_fusion.append(Gate(input_size, hidden_size=128, embed_size=embed_dim))
                    else:
                        raise NameError
                else:
                    self.seg_input_fusion.append(None)
        self.mem_lstm = nn.ModuleList()
        for _ in range(n_blocks - 1):
            self.mem_lstm.append(MemLSTM(hidden_size, causal=causal, dropout=dropout))
        self.output_fc = nn.Sequential(nn.PReLU(), nn.Conv1d(input_size, output_size, 1))

    def _build(self, device):
        if self.output_fc[0].weight.numel() != 1:
            raise NotImplementedError("PReLU with per-channel slopes is not on the HIP path")
        return dict(out=linear_plan(self.output_fc[1], device), out_slope=_f32(self.output_fc[0].weight, device))

    def padded_frames_needed(self, t: int) -> int:
        """T' = T + rest, rest = K - T % K in [1, K] (skim.py:429-433); overlapped segments get their own buffer."""
        return t if self.seg_overlap else t + self.seg_size - t % self.seg_size

    def _fuse(self, i: int, x: torch.Tensor, tp: int, embed: Optional[torch.Tensor]) -> torch.Tensor:
        if embed is not None and self.block_with_embed[i]:
            return self.seg_input_fusion[i].forward_padded(x, tp, embed, self.embed_norm)
        return x

    def _output(self, x: torch.Tensor, t: int, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        p = self._plan_get(x.device, self._build)
        pro = hip.make_prologue(0, True, None, 0.0, 0.0, None, None, p["out_slope"])
        if out is None:
            out = torch.empty(x.shape[0], p["out"]["M"], x.shape[2], dtype=torch.float32, device=x.device)
        elif tuple(out.shape) != (x.shape[0], p["out"]["M"], x.shape[2]) or not out.is_contiguous():
            raise ValueError(f"_output: `out` must be a contiguous {(x.shape[0], p['out']['M'], x.shape[2])} tensor")
        y, _ = hip.conv1x1(x, t, p["out"]["wt"], p["out"]["M"], pro, p["out"]["bias"], out=out)
        return y

    def forward_padded(self, x_pad: torch.Tensor, t: int, embed: Optional[torch.Tensor] = None,
                       lane: int = 0) -> torch.Tensor:
        """padded [N,C,ldt] (zero beyond T, ldt >= padded_frames_needed(T)), embed [N,E] -> mask logits padded."""
        if self.seg_overlap:
            x_pad, tp = hip.segment_split(x_pad, t, self.seg_size)     # SkiM.split (skim.py:334-369)
        else:
            tp = self.padded_frames_needed(t)
        if x_pad.shape[-1] < tp:
            raise RuntimeError(f"SkiM: rows hold {x_pad.shape[-1]} frames, the segment padding needs {tp}")
        k = self.seg_size
        s = tp // k
        x = x_pad
        h = c = None
        shift = 0
        for i in range(self.n_blocks):
            x = self._fuse(i, x, tp, embed)
            x, (h, c) = self.seg_lstm[i].forward_padded(x, tp, s, k, h, c, shift)
            if i < self.n_blocks - 1:
                h, c, _, _ = self.mem_lstm[i].forward_state(h, c, s)
                shift = 1 if self.causal else 0
        if self.seg_overlap:
            x = hip.segment_merge(x, tp, t, self.seg_size)              # SkiM.merge (skim.py:371-408)
        return self._output(x, t)

    def forward(self, x: torch.Tensor, embed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [N,C,T], embed [N,E] -> [N,C_out,T] (skim.py:410-469)."""
        hip.require_device(x, "SkiM.forward")
        t = x.shape[-1]
        return hip.unpad_rows(self.forward_padded(hip.pad_rows(x, self.padded_frames_needed(t)), t, embed), t)
