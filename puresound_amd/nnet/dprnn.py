"""Dual-path RNN masker on the HIP path (mirror of puresound/nnet/dprnn.py:10-244).

Layout: the padded channel-major rows [N][C][ldt] of the rest of the library, holding T' = S*K frames (the
reference pads T up to the next multiple of K and, when T % K == 0, by one whole extra segment; dprnn.py:142-147).
A segment is K consecutive frames, so the reference's [N,S,K,C] <-> [N,K,S,C] transposes become two addressing
modes of ps_lstm_f32 (intra: sequences of K contiguous frames; inter: sequences of S frames K apart) and no data
moves.  Every Linear is a ps_conv1x1_f32; LayerNorm + residual is ps_chan_layernorm_f32.
"""
from typing import List, Optional

import torch
import torch.nn as nn

from .. import hip
from ..ops import op_module
from ._plans import PlanCache, _f32, layernorm_plan, linear_plan, lstm_path, lstm_plan
from .lobe.trivial import FiLM


def _out_size(ctor, x, aux, params):
    return (x[0], ctor["output_size"], x[2])


@op_module("dprnn_fwd", _out_size)
class DPRNN(PlanCache, nn.Module):
    """Deep dual-path RNN (dprnn.py:10-109); constructor order as the reference (dprnn.py:27-40)."""

    def __init__(self, input_size: int, hidden_size: int, output_size: int, n_blocks: int = 2, seg_size: int = 20,
                 seg_overlap: bool = False, causal: bool = True, embed_dim: int = 0, embed_norm: bool = False,
                 block_with_embed: Optional[List] = None, embedding_free_tse: bool = False):
        super().__init__()
        self.seg_size = seg_size
        self.seg_overlap = seg_overlap
        self.input_size = input_size
        self.hidden_size = hidden_size
        self.bi_direct = not causal
        self.n_blocks = n_blocks
        self.embed_dim = embed_dim
        self.embed_norm = embed_norm
        self.block_with_embed = block_with_embed
        self.embedding_free_tse = embedding_free_tse
        self._last_amax = None

        self.input_film = nn.ModuleList()
        self.intra_rnn = nn.ModuleList()
        self.intra_proj = nn.ModuleList()
        self.intra_norm = nn.ModuleList()
        self.inter_rnn = nn.ModuleList()
        self.inter_norm = nn.ModuleList()
        self.inter_proj = nn.ModuleList()
        d = int(self.bi_direct) + 1
        for i in range(n_blocks):
            self.intra_rnn.append(nn.LSTM(input_size, hidden_size, num_layers=1, bidirectional=self.bi_direct,
                                          batch_first=True))
            if embed_dim != 0 and block_with_embed[i]:
                self.input_film.append(FiLM(input_size, embed_dim, input_norm=True))
            else:
                self.input_film.append(None)
            self.intra_proj.append(nn.Linear(hidden_size * d, input_size))
            self.intra_norm.append(nn.LayerNorm(input_size))
            self.inter_rnn.append(nn.LSTM(input_size, hidden_size, num_layers=1, bidirectional=self.bi_direct,
                                          batch_first=True))
            self.inter_proj.append(nn.Linear(hidden_size * d, input_size))
            self.inter_norm.append(nn.LayerNorm(input_size))
        self.output_fc = nn.Sequential(nn.PReLU(), nn.Conv1d(input_size, output_size, 1))

    # -- plan ---------------------------------------------------------------------------------------
    def _build(self, device):
        blocks = []
        for i in range(self.n_blocks):
            blocks.append(dict(
                intra=(lstm_plan(self.intra_rnn[i], device, self.gemm_precision), linear_plan(self.intra_proj[i], device),
                       layernorm_plan(self.intra_norm[i], device)),
                inter=(lstm_plan(self.inter_rnn[i], device, self.gemm_precision), linear_plan(self.inter_proj[i], device),
                       layernorm_plan(self.inter_norm[i], device))))
        if self.output_fc[0].weight.numel() != 1:
            raise NotImplementedError("PReLU with per-channel slopes is not on the HIP path")
        plan = dict(blocks=blocks, out=linear_plan(self.output_fc[1], device),
                    out_slope=_f32(self.output_fc[0].weight, device))
        if self.gemm_precision == "fp16x2" and self.output_fc[1].in_channels >= 64:
            # output conv in the fp16x2 arithmetic too (its input range comes from the last row kernel); |slope| <= 1 keeps
            # PReLU(x) inside the range measured for x (read once, at plan time)
            if abs(float(self.output_fc[0].weight.detach().reshape(-1)[0])) <= 1.0:
                plan["out_f16x2"] = hip.pack_wt_f16x2(_f32(self.output_fc[1].weight[:, :, 0], device))
        return plan

    # -- segment geometry -------------------------------------------------------------------------------
    def padded_frames_needed(self, t: int) -> int:
        """T' = T + rest, rest = K - T % K in [1, K] (dprnn.py:142-147); overlapped segments are re-laid out
        into a buffer of their own, so the input rows only need their T frames."""
        return t if self.seg_overlap else t + self.seg_size - t % self.seg_size

    def _run_blocks(self, x: torch.Tensor, tp: int, embed, init_states, want_states: bool):
        p = self._plan_get(x.device, self._build)
        k = self.seg_size
        s = tp // k
        states = []
        amax = [None]   # fp16x2 input projections: every recurrence hands the next one the range of its output
        for i, blk in enumerate(p["blocks"]):
            if embed is not None and self.block_with_embed[i]:
                x = self.input_film[i].forward_padded(x, tp, embed, self.embed_norm)
                amax[0] = None
            x, _ = lstm_path(x, tp, *blk["intra"], q=s, q_stride=k, steps=k, step_stride=1, amax=amax)
            h0, c0 = init_states[i] if init_states is not None else (None, None)
            x, st = lstm_path(x, tp, *blk["inter"], q=k, q_stride=1, steps=s, step_stride=k, h0=h0, c0=c0,
                              want_state=want_states, amax=amax)
            states.append(st)
        self._last_amax = amax[0]   # (partial maxima of the block stack's output, or None; read by forward_padded only)
        return x, states

    def hidden_states_padded(self, e_pad: torch.Tensor, te: int):
        """DPRNN._get_hidden_states (dprnn.py:193-244): final inter-LSTM states of an enrolment pass, per block,
        in state layout [N, D*H, ldq]."""
        if self.seg_overlap:
            e_pad, tp = hip.segment_split(e_pad, te, self.seg_size)
        else:
            tp = self.padded_frames_needed(te)
        if e_pad.shape[-1] < tp:
            raise RuntimeError("DPRNN: enrolment rows are shorter than the segment padding")
        _, states = self._run_blocks(e_pad, tp, None, None, True)
        return states

    def forward_padded(self, x_pad: torch.Tensor, t: int, embed: Optional[torch.Tensor] = None, lane: int = 0,
                       embed_frames: Optional[int] = None) -> torch.Tensor:
        """padded [N,C,ldt] (zero beyond T, ldt >= padded_frames_needed(T)) -> mask logits padded [N,C_out,ldt].
        embed: [N,E] vector, or in embedding-free mode the padded enrolment features with `embed_frames` frames."""
        if self.seg_overlap:
            x_pad, tp = hip.segment_split(x_pad, t, self.seg_size)     # SplitMerge.split (dprnn.py:136-139)
        else:
            tp = self.padded_frames_needed(t)
        if x_pad.shape[-1] < tp:
            raise RuntimeError(f"DPRNN: rows hold {x_pad.shape[-1]} frames, the segment padding needs {tp}")
        init = None
        if self.embedding_free_tse:
            # the reference asserts on embed.dim() == 3 (dprnn.py:121-124)
            assert embed is not None and embed.dim() == 3, "embedding free tse need enrollment waveform as input."
            init = self.hidden_states_padded(embed, embed_frames)
            embed = None
        x, _ = self._run_blocks(x_pad, tp, embed, init, False)
        x_amax = self._last_amax
        self._last_amax = None
        if self.seg_overlap:
            x = hip.segment_merge(x, tp, t, self.seg_size)              # SplitMerge.merge (dprnn.py:182-185)
            x_amax = None
        p = self._plan
        pro = hip.make_prologue(0, True, None, 0.0, 0.0, None, None, p["out_slope"])
        if "out_f16x2" in p:
            wf, we = p["out_f16x2"]
            # (the maxima cover the tp >= t frames the blocks ran on: a bound for the first t)
            y, _, _ = hip.conv1x1_f16x2(x, t, wf, we, p["out"]["M"], pro, p["out"]["bias"],
                                        x_amax=x_amax if x_amax is not None else hip.absmax(x, t))
            return y
        y, _ = hip.conv1x1(x, t, p["out"]["wt"], p["out"]["M"], pro, p["out"]["bias"])
        return y

    def forward(self, x: torch.Tensor, embed: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [N,C,T], embed [N,E] (or [N,C,T_e] enrolment features when embedding_free_tse) -> [N,C_out,T]
        (dprnn.py:111-191)."""
        hip.require_device(x, "DPRNN.forward")
        t = x.shape[-1]
        x_pad = hip.pad_rows(x, self.padded_frames_needed(t))
        frames = None
        if self.embedding_free_tse and embed is not None and embed.dim() == 3:
            frames = embed.shape[-1]
            embed = hip.pad_rows(embed, self.padded_frames_needed(frames))
        return hip.unpad_rows(self.forward_padded(x_pad, t, embed, embed_frames=frames), t)
