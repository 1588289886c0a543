"""CPU oracle for the recurrent maskers (DPRNN, SkiM, StreamingSkiM) -- TEST INFRASTRUCTURE, NOT PRODUCT.

Plain tensor restatement of puresound/nnet/dprnn.py, puresound/nnet/skim.py, puresound/streaming/
skim_inference.py and the FiLM / Gate / SplitMerge lobes they use (puresound/nnet/lobe/trivial.py) of
mcw519/PureSound.  Same conventions as separator_oracle.py: state_dict driven (flat {key: tensor} + prefix),
arithmetic in the dtype of the inputs, no nn.LSTM / nn.LayerNorm / F.conv1d calls -- the LSTM cell, the layer
norm and the 1x1 convolutions are written out.  Only tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg may import this file.

Pinning: tests/golden/make_golden.py runs the imported reference on formula-generated weights and stores the
vectors this file is checked against (tests/test_oracle_golden.py).  The reference's own tests pin only
streaming == offline (tests/test_streaming.py:10-116) and split/merge identity (tests/test_lobe.py:50-54);
both properties are re-checked on this oracle as well.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Tuple

import torch

from .separator_oracle import chan_ln, conv1x1, prelu

SD = Dict[str, torch.Tensor]
States = Tuple[torch.Tensor, torch.Tensor]


# ---------------------------------------------------------------------------
# elementary pieces
# ---------------------------------------------------------------------------
def layer_norm(x: torch.Tensor, w: torch.Tensor, b: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    """nn.LayerNorm(C) over the last dim (biased variance, eps inside the sqrt)."""
    mean = x.mean(-1, keepdim=True)
    var = ((x - mean) ** 2).mean(-1, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * w.to(x.dtype) + b.to(x.dtype)


def linear(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor] = None) -> torch.Tensor:
    y = torch.matmul(x, w.to(x.dtype).t())
    return y if b is None else y + b.to(x.dtype)


def lstm_direction(x: torch.Tensor, w_ih, w_hh, b_ih, b_hh, h0: torch.Tensor, c0: torch.Tensor,
                   reverse: bool) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor]:
    """One direction of a 1-layer nn.LSTM(batch_first=True): x [B,L,I] -> out [B,L,H], h_L, c_L.
    Gate order i, f, g, o; c' = sig(f) c + sig(i) tanh(g); h' = sig(o) tanh(c')."""
    hid = w_hh.shape[1]
    gx = linear(x, w_ih, b_ih.to(x.dtype) + b_hh.to(x.dtype))  # [B,L,4H]
    h, c = h0, c0
    out = torch.empty(x.shape[0], x.shape[1], hid, dtype=x.dtype)
    steps = range(x.shape[1] - 1, -1, -1) if reverse else range(x.shape[1])
    for t in steps:
        a = gx[:, t] + linear(h, w_hh)
        i, f = torch.sigmoid(a[:, :hid]), torch.sigmoid(a[:, hid:2 * hid])
        g, o = torch.tanh(a[:, 2 * hid:3 * hid]), torch.sigmoid(a[:, 3 * hid:])
        c = f * c + i * g
        h = o * torch.tanh(c)
        out[:, t] = h
    return out, h, c


def lstm(x: torch.Tensor, sd: SD, p: str, bidirectional: bool, states: Optional[States] = None):
    """nn.LSTM(I, H, num_layers=1, bidirectional, batch_first=True): x [B,L,I] -> out [B,L,D*H],
    (h_n [D,B,H], c_n [D,B,H])."""
    dirs = ["", "_reverse"] if bidirectional else [""]
    hid = sd[p + "weight_hh_l0"].shape[1]
    outs, hs, cs = [], [], []
    for d, suf in enumerate(dirs):
        if states is None:
            h0 = torch.zeros(x.shape[0], hid, dtype=x.dtype)
            c0 = torch.zeros(x.shape[0], hid, dtype=x.dtype)
        else:
            h0, c0 = states[0][d].to(x.dtype), states[1][d].to(x.dtype)
        o, h, c = lstm_direction(x, sd[f"{p}weight_ih_l0{suf}"], sd[f"{p}weight_hh_l0{suf}"],
                                 sd[f"{p}bias_ih_l0{suf}"], sd[f"{p}bias_hh_l0{suf}"], h0, c0, d == 1)
        outs.append(o)
        hs.append(h)
        cs.append(c)
    return torch.cat(outs, -1), (torch.stack(hs), torch.stack(cs))


def gru_rnn(x: torch.Tensor, sd: SD, p: str, bidirectional: bool, kind: str) -> torch.Tensor:
    """nn.GRU / nn.RNN(I, H, num_layers=1, bidirectional, batch_first=True) as SingleRNN builds them for rnn_type "GRU" / "RNN"
    (lobe/rnn.py:19-35 of mcw519/PureSound: getattr(nn, rnn_type), default tanh non-linearity): x [B,L,I] -> out [B,L,D*H].
    GRU (gate order r, z, n):  r = sig(W_ir x + b_ir + W_hr h + b_hr),  z = sig(W_iz x + b_iz + W_hz h + b_hz),
    n = tanh(W_in x + b_in + r * (W_hn h + b_hn)),  h' = (1 - z) * n + z * h.      RNN:  h' = tanh(W_ih x + b_ih + W_hh h + b_hh)."""
    dirs = ["", "_reverse"] if bidirectional else [""]
    hid = sd[p + "weight_hh_l0"].shape[1]
    outs = []
    for d, suf in enumerate(dirs):
        w_ih, w_hh = sd[f"{p}weight_ih_l0{suf}"], sd[f"{p}weight_hh_l0{suf}"]
        b_ih, b_hh = sd[f"{p}bias_ih_l0{suf}"].to(x.dtype), sd[f"{p}bias_hh_l0{suf}"].to(x.dtype)
        gx = linear(x, w_ih, b_ih)
        h = torch.zeros(x.shape[0], hid, dtype=x.dtype)
        out = torch.empty(x.shape[0], x.shape[1], hid, dtype=x.dtype)
        steps = range(x.shape[1] - 1, -1, -1) if d == 1 else range(x.shape[1])
        for t in steps:
            gh = linear(h, w_hh, b_hh)
            if kind == "GRU":
                r = torch.sigmoid(gx[:, t, :hid] + gh[:, :hid])
                z = torch.sigmoid(gx[:, t, hid:2 * hid] + gh[:, hid:2 * hid])
                n = torch.tanh(gx[:, t, 2 * hid:] + r * gh[:, 2 * hid:])
                h = (1 - z) * n + z * h
            else:
                h = torch.tanh(gx[:, t] + gh)
            out[:, t] = h
        outs.append(out)
    return torch.cat(outs, -1)


def l2_normalize(e: torch.Tensor) -> torch.Tensor:
    """F.normalize(e, p=2, dim=1)."""
    return e / e.norm(p=2, dim=1, keepdim=True).clamp_min(1e-12)


def film(x: torch.Tensor, cond: torch.Tensor, sd: SD, p: str, input_norm: bool = True) -> torch.Tensor:
    """FiLM.forward (lobe/trivial.py:148-167): x [N,C,T], cond [N,E] -> scale(x;c) * x + bias(x;c)."""
    if input_norm:
        x = layer_norm(x.transpose(1, 2), sd[p + "norm.weight"], sd[p + "norm.bias"]).transpose(1, 2)
    c = torch.cat([x, cond.to(x.dtype).unsqueeze(-1).expand(-1, -1, x.shape[-1])], dim=1)
    return conv1x1(c, sd[p + "cond_scale.weight"]) * x + conv1x1(c, sd[p + "cond_bias.weight"])


def gate(x: torch.Tensor, cond: torch.Tensor, sd: SD, p: str) -> torch.Tensor:
    """Gate.forward (lobe/trivial.py:61-126): 1x1 in_conv; left = PReLU(cLN(1x1)); right =
    sigmoid(PReLU(cLN(1x1 on [x; cond]))); out_conv(left * right) + x."""
    h = conv1x1(x, sd[p + "in_conv.weight"])
    h_r = torch.cat([h, cond.to(x.dtype).unsqueeze(-1).expand(-1, -1, h.shape[-1])], dim=1)
    left = conv1x1(h, sd[p + "left_conv.0.weight"])
    left = prelu(chan_ln(left, sd[p + "left_conv.1.gamma"], sd[p + "left_conv.1.beta"]), sd[p + "left_conv.2.weight"])
    right = conv1x1(h_r, sd[p + "right_conv.0.weight"])
    right = prelu(chan_ln(right, sd[p + "right_conv.1.gamma"], sd[p + "right_conv.1.beta"]),
                  sd[p + "right_conv.2.weight"])
    return conv1x1(left * torch.sigmoid(right), sd[p + "out_conv.weight"]) + x


def split_overlap(x: torch.Tensor, seg_size: int) -> Tuple[torch.Tensor, int]:
    """SplitMerge.split / SkiM.split (lobe/trivial.py:178-214): 50 % overlapped segments.
    x [N,C,T] -> [N,S,K,C], rest."""
    stride = seg_size // 2
    n, c, t = x.shape
    rest = seg_size - (stride + t % seg_size) % seg_size
    xp = torch.zeros(n, c, stride + t + rest + stride, dtype=x.dtype)
    xp[:, :, stride:stride + t] = x
    total = xp.shape[-1]
    n_half = (total - stride) // seg_size          # segments of each of the two interleaved streams
    seg = torch.empty(n, c, 2 * n_half, seg_size, dtype=x.dtype)
    for j in range(n_half):
        seg[:, :, 2 * j] = xp[:, :, j * seg_size:(j + 1) * seg_size]
        seg[:, :, 2 * j + 1] = xp[:, :, stride + j * seg_size:stride + (j + 1) * seg_size]
    return seg.permute(0, 2, 3, 1), rest


def merge_overlap(x: torch.Tensor, rest: int) -> torch.Tensor:
    """SplitMerge.merge (lobe/trivial.py:216-241): [N,S,K,C] -> [N,C,T], mean of the two covers."""
    n, s, k, c = x.shape
    stride = k // 2
    xs = x.permute(0, 3, 1, 2)                      # [N,C,S,K]
    a = xs[:, :, 0::2].reshape(n, c, -1)[:, :, stride:]
    b = xs[:, :, 1::2].reshape(n, c, -1)[:, :, :-stride]
    out = (a + b) / 2
    return out[..., :-rest] if rest > 0 else out


def segment(x: torch.Tensor, seg_size: int, seg_overlap: bool):
    """[N,C,T] -> ([N,S,K,C], rest).  Without overlap the reference ALWAYS pads (rest = K - T % K is never 0:
    dprnn.py:142-147, skim.py:429-433)."""
    if seg_overlap:
        return split_overlap(x, seg_size)
    n, c, t = x.shape
    rest = seg_size - t % seg_size
    xp = torch.zeros(n, t + rest, c, dtype=x.dtype)
    xp[:, :t] = x.transpose(1, 2)
    return xp.reshape(n, -1, seg_size, c), rest


def output_fc(x: torch.Tensor, sd: SD, p: str) -> torch.Tensor:
    """nn.Sequential(PReLU(), Conv1d(C, C_out, 1)) on [N,C,T]."""
    return conv1x1(prelu(x, sd[p + "0.weight"]), sd[p + "1.weight"], sd[p + "1.bias"])


# ---------------------------------------------------------------------------
# DPRNN (dprnn.py:111-244)
# ---------------------------------------------------------------------------
def _dprnn_blocks(x4: torch.Tensor, sd: SD, p: str, args: dict, embed_rep: Optional[torch.Tensor],
                  init_states: List[Optional[States]]):
    n, s, k, c = x4.shape
    bi = not args["causal"]
    out = x4
    hidden = []
    for i in range(args["n_blocks"]):
        out = out.reshape(-1, k, c)
        if embed_rep is not None and args["block_with_embed"][i]:
            out = film(out.transpose(1, 2), embed_rep, sd, f"{p}input_film.{i}.").transpose(1, 2)
        y, _ = lstm(out, sd, f"{p}intra_rnn.{i}.", bi)
        y = linear(y, sd[f"{p}intra_proj.{i}.weight"], sd[f"{p}intra_proj.{i}.bias"])
        out = out + layer_norm(y, sd[f"{p}intra_norm.{i}.weight"], sd[f"{p}intra_norm.{i}.bias"])
        inter_in = out.reshape(n, s, k, c).permute(0, 2, 1, 3).reshape(-1, s, c)
        y, hid = lstm(inter_in, sd, f"{p}inter_rnn.{i}.", bi, init_states[i])
        hidden.append(hid)
        y = linear(y, sd[f"{p}inter_proj.{i}.weight"], sd[f"{p}inter_proj.{i}.bias"])
        out = inter_in + layer_norm(y, sd[f"{p}inter_norm.{i}.weight"], sd[f"{p}inter_norm.{i}.bias"])
        out = out.reshape(n, k, s, c).permute(0, 2, 1, 3)
    return out, hidden


def dprnn_hidden_states(x: torch.Tensor, sd: SD, p: str, args: dict) -> List[States]:
    """DPRNN._get_hidden_states (dprnn.py:193-244): the enrolment pass of the embedding-free TSE mode."""
    x4, _ = segment(x, args["seg_size"], args["seg_overlap"])
    _, hidden = _dprnn_blocks(x4, sd, p, args, None, [None] * args["n_blocks"])
    return hidden


def dprnn(x: torch.Tensor, sd: SD, p: str, args: dict, embed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """DPRNN.forward (dprnn.py:111-191).  args: n_blocks, seg_size, seg_overlap, causal, embed_norm,
    block_with_embed, embedding_free_tse."""
    if args.get("embedding_free_tse", False):
        assert embed is not None and embed.dim() == 3, "embedding free tse need enrollment waveform as input."
        init = dprnn_hidden_states(embed, sd, p, args)
    else:
        init = [None] * args["n_blocks"]
        if args.get("embed_norm", False) and embed is not None:
            embed = l2_normalize(embed)
    t = x.shape[-1]
    x4, rest = segment(x, args["seg_size"], args["seg_overlap"])
    n, s, k, c = x4.shape
    rep = None
    if not args.get("embedding_free_tse", False) and embed is not None:
        rep = embed.unsqueeze(1).expand(-1, s, -1).reshape(n * s, -1)
    out, _ = _dprnn_blocks(x4, sd, p, args, rep, init)
    if args["seg_overlap"]:
        y = merge_overlap(out, rest)
    else:
        y = out.reshape(n, s * k, c)[:, :t].transpose(1, 2)
    return output_fc(y, sd, p + "output_fc.")


# ---------------------------------------------------------------------------
# SkiM (skim.py)
# ---------------------------------------------------------------------------
def seg_lstm(x: torch.Tensor, h: Optional[torch.Tensor], c: Optional[torch.Tensor], sd: SD, p: str, causal: bool):
    """SegLSTM.forward (skim.py:198-229): x [B,K,C], h/c [D,B,H] -> x + LN(proj(LSTM(x))), h, c."""
    hid = sd[p + "lstm.weight_hh_l0"].shape[1]
    d = 1 if causal else 2
    if h is None:
        h = torch.zeros(d, x.shape[0], hid, dtype=x.dtype)
    if c is None:
        c = torch.zeros(d, x.shape[0], hid, dtype=x.dtype)
    y, (h, c) = lstm(x, sd, p + "lstm.", not causal, (h, c))
    y = linear(y, sd[p + "proj.weight"], sd[p + "proj.bias"])
    return x + layer_norm(y, sd[p + "norm.weight"], sd[p + "norm.bias"]), h, c


def mem_lstm(h: torch.Tensor, c: torch.Tensor, sd: SD, p: str, causal: bool, h_states: Optional[States] = None,
             c_states: Optional[States] = None, streaming: bool = False):
    """MemLSTM.forward (skim.py:45-114): h, c [N,S,D,H] -> next-block initial states [D,N*S,H] (+ LSTM states)."""
    n, s, d, hid = h.shape
    outs, states = [], []
    for name, v, st in (("h", h, h_states), ("c", c, c_states)):
        v = v.reshape(n, s, -1)
        y, new_st = lstm(v, sd, f"{p}{name}_net.", not causal, st)
        y = linear(y, sd[f"{p}{name}_proj.weight"], sd[f"{p}{name}_proj.bias"])
        v = v + layer_norm(y, sd[f"{p}{name}_norm.weight"], sd[f"{p}{name}_norm.bias"])
        v = v.reshape(n * s, d, hid).transpose(1, 0)  # [D, NS, H]
        if causal and not streaming:
            # skim.py:102-109 shifts by one along the flattened N*S axis, so with N > 1 the first segment of
            # utterance n receives the last segment's state of utterance n-1 (kept as the reference does it)
            shifted = torch.zeros_like(v)
            shifted[:, 1:] = v[:, :-1]
            v = shifted
        outs.append(v)
        states.append(new_st)
    return outs[0], outs[1], states[0], states[1]


def fusion(x: torch.Tensor, cond: torch.Tensor, sd: SD, p: str, kind: str) -> torch.Tensor:
    k = kind.lower()
    if k == "film":
        return film(x, cond, sd, p)
    if k == "gate":
        return gate(x, cond, sd, p)
    raise NameError


def skim(x: torch.Tensor, sd: SD, p: str, args: dict, embed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """SkiM.forward (skim.py:410-469).  args: n_blocks, seg_size, seg_overlap, causal, embed_norm, embed_fusion,
    block_with_embed, hidden_size."""
    if args.get("embed_norm", False) and embed is not None:
        embed = l2_normalize(embed)
    t = x.shape[-1]
    x4, rest = segment(x, args["seg_size"], args["seg_overlap"])
    n, s, k, c = x4.shape
    rep = None if embed is None else embed.unsqueeze(1).expand(-1, s, -1).reshape(n * s, -1)
    out = x4.reshape(n * s, k, c)
    h = cst = None
    hid = args["hidden_size"]
    for i in range(args["n_blocks"]):
        if rep is not None and args["block_with_embed"][i]:
            out = fusion(out.transpose(1, 2), rep, sd, f"{p}seg_input_fusion.{i}.", args["embed_fusion"]).transpose(1, 2)
        out, h, cst = seg_lstm(out, h, cst, sd, f"{p}seg_lstm.{i}.", args["causal"])
        if i < args["n_blocks"] - 1:
            h4 = h.reshape(-1, n, s, hid).permute(1, 2, 0, 3)
            c4 = cst.reshape(-1, n, s, hid).permute(1, 2, 0, 3)
            h, cst, _, _ = mem_lstm(h4, c4, sd, f"{p}mem_lstm.{i}.", args["causal"])
    if args["seg_overlap"]:
        y = merge_overlap(out.reshape(n, s, k, c), rest)
    else:
        y = out.reshape(n, s * k, c)[:, :t].transpose(1, 2)
    return output_fc(y, sd, p + "output_fc.")


# ---------------------------------------------------------------------------
# StreamingSkiM (streaming/skim_inference.py)
# ---------------------------------------------------------------------------
class SkimStream:
    """Frame API state machine of StreamingSkiM (init_status / step_frame / update_mem_lstm /
    reset_seg_lstm_status, skim_inference.py:142-252) for B independent streams at once: every state
    tensor carries a stream axis where the reference has the constant 1."""

    def __init__(self, sd: SD, p: str, args: dict, streams: int = 1, dtype=torch.float32):
        self.sd, self.p, self.args, self.b = sd, p, args, streams
        d = 1 if args["causal"] else 2
        hid, nb = args["hidden_size"], args["n_blocks"]
        z = lambda: torch.zeros(d, streams, hid, dtype=dtype)  # noqa: E731
        self.frames_counter = 0
        self.seg_h = [z() for _ in range(nb)]
        self.seg_c = [z() for _ in range(nb)]
        self.mem_h_hidden = [(z(), z()) for _ in range(nb - 1)]
        self.mem_c_hidden = [(z(), z()) for _ in range(nb - 1)]

    def step_frame(self, x: torch.Tensor, embed: Optional[torch.Tensor]) -> torch.Tensor:
        """x [B,1,C], embed [B,E] -> [B,C_out,1] (skim_inference.py:176-218)."""
        a, sd, p = self.args, self.sd, self.p
        if a.get("embed_norm", False) and embed is not None:
            embed = l2_normalize(embed)
        for i in range(a["n_blocks"]):
            if embed is not None and a["block_with_embed"][i]:
                x = fusion(x.transpose(1, 2), embed, sd, f"{p}seg_input_fusion.{i}.", a["embed_fusion"]).transpose(1, 2)
            x, self.seg_h[i], self.seg_c[i] = seg_lstm(x, self.seg_h[i], self.seg_c[i], sd, f"{p}seg_lstm.{i}.",
                                                       a["causal"])
        out = output_fc(x.transpose(1, 2), sd, p + "output_fc.")
        self.frames_counter += 1
        if self.frames_counter % a["seg_size"] == 0:
            self.update_mem_lstm()
            self.seg_h[0] = torch.zeros_like(self.seg_h[0])
            self.seg_c[0] = torch.zeros_like(self.seg_c[0])
            self.frames_counter = 0
        return out

    def update_mem_lstm(self) -> None:
        """skim_inference.py:220-252: block i's segment-end state -> MemLSTM i -> block i+1's initial state."""
        a = self.args
        hid = a["hidden_size"]
        cur_h = [t.clone() for t in self.seg_h]
        cur_c = [t.clone() for t in self.seg_c]
        for i in range(a["n_blocks"] - 1):
            # [D,B,H] -> [B,1,D,H]: every stream is its own batch entry with one segment
            h4 = cur_h[i].permute(1, 0, 2).unsqueeze(1)
            c4 = cur_c[i].permute(1, 0, 2).unsqueeze(1)
            mh, mc, hs, cs = mem_lstm(h4, c4, self.sd, f"{self.p}mem_lstm.{i}.", a["causal"], self.mem_h_hidden[i],
                                      self.mem_c_hidden[i], streaming=True)
            self.seg_h[i + 1], self.seg_c[i + 1] = mh, mc
            self.mem_h_hidden[i], self.mem_c_hidden[i] = hs, cs


def skim_step_chunk(x: torch.Tensor, sd: SD, p: str, args: dict, seg_h, mem_h_hidden, seg_c, mem_c_hidden,
                    embed: Optional[torch.Tensor] = None):
    """StreamingSkiM.step_chunk (skim_inference.py:41-139): x [B,K,C] = one whole segment; returns
    (out [B,C_out,K], seg_h[:-1], mem_h_hidden, seg_c[:-1], mem_c_hidden)."""
    a = args
    nb = a["n_blocks"]
    if a.get("embed_norm", False) and embed is not None:
        embed = l2_normalize(embed)
    if seg_h is not None and seg_c is not None:
        seg_h = [None] + [seg_h[i] for i in range(nb - 1)]
        seg_c = [None] + [seg_c[i] for i in range(nb - 1)]
    else:
        seg_h, seg_c = [None] * nb, [None] * nb
    if mem_h_hidden is None and mem_c_hidden is None:
        mem_h_hidden, mem_c_hidden = [None] * (nb - 1), [None] * (nb - 1)
    else:
        mem_h_hidden, mem_c_hidden = list(mem_h_hidden), list(mem_c_hidden)
    outs = []
    for f in range(x.shape[1]):
        cur = x[:, f:f + 1]
        for i in range(nb):
            if embed is not None and a["block_with_embed"][i]:
                cur = fusion(cur.transpose(1, 2), embed, sd, f"{p}seg_input_fusion.{i}.", a["embed_fusion"]).transpose(1, 2)
            cur, seg_h[i], seg_c[i] = seg_lstm(cur, seg_h[i], seg_c[i], sd, f"{p}seg_lstm.{i}.", a["causal"])
        outs.append(output_fc(cur.transpose(1, 2), sd, p + "output_fc."))
    for i in range(nb - 1):
        h4 = seg_h[i].permute(1, 0, 2).unsqueeze(1)
        c4 = seg_c[i].permute(1, 0, 2).unsqueeze(1)
        mh, mc, hs, cs = mem_lstm(h4, c4, sd, f"{p}mem_lstm.{i}.", a["causal"], mem_h_hidden[i], mem_c_hidden[i],
                                  streaming=True)
        seg_h[i], seg_c[i] = mh, mc
        mem_h_hidden[i], mem_c_hidden[i] = hs, cs
    return torch.cat(outs, -1), seg_h[:-1], mem_h_hidden, seg_c[:-1], mem_c_hidden


def overlap_add_mean(a: Optional[torch.Tensor], b: torch.Tensor, overlap: int) -> torch.Tensor:
    """egs/tse/demo/utils.py:121-128 on [B, L] rows: the overlapped samples are AVERAGED."""
    if a is None:
        return b
    return torch.cat([a[..., :-overlap], (a[..., -overlap:] + b[..., :overlap]) / 2, b[..., overlap:]], dim=-1)


class DemoStream:
    """DemoTseNet.streaming_inference(_chunk) (egs/tse/demo/utils.py:78-119) for B streams: 32-sample sliding
    window, encoder, one masker frame step, mask multiply, decoder, averaging 16-sample overlap-add."""

    def __init__(self, sd: SD, args: dict, streams: int, win: int = 32, hop: int = 16):
        self.sd, self.win, self.hop = sd, win, hop
        self.masker = SkimStream(sd, "masker.", args, streams)
        self.queue = None

    def step_hop(self, hop_samples: torch.Tensor, embed: torch.Tensor) -> Optional[torch.Tensor]:
        if self.queue is None:
            self.queue = torch.cat([torch.zeros_like(hop_samples), hop_samples], -1)
            return None
        self.queue = torch.cat([self.queue[:, self.hop:], hop_samples], -1)
        w_enc, w_dec = self.sd["encoder.encoder.weight"][:, 0], self.sd["encoder.decoder.weight"][:, 0]
        feats = torch.relu(self.queue @ w_enc.t())                       # FreeEncDec(output_active=True), one frame
        mask = self.masker.step_frame(feats.unsqueeze(1), embed)[..., 0]  # [B,C]
        return (feats * mask) @ w_dec                                    # ConvTranspose1d of a single frame

    def step_chunk(self, chunk: torch.Tensor, embed: torch.Tensor, pre: Optional[torch.Tensor] = None):
        for i in range(chunk.shape[-1] // self.hop):
            cur = self.step_hop(chunk[:, i * self.hop:(i + 1) * self.hop], embed)
            if cur is not None:
                pre = overlap_add_mean(pre, cur, self.win - self.hop)
        return pre
