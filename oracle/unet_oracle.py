"""CPU oracle for the 2-D convolutional maskers (Unet, UnetTcn, DPCRN) -- TEST INFRASTRUCTURE, NOT PRODUCT.

Plain tensor restatement of puresound/nnet/unet.py:13-557 and puresound/nnet/dpcrn.py:11-213 of mcw519/PureSound
(SURVEY section 8(f) rows 1-2).  Same conventions as separator_oracle.py: state_dict driven, arithmetic in the dtype of
the inputs, no F.conv2d / F.conv_transpose2d / nn.LSTM / nn.LayerNorm / nn.BatchNorm2d calls -- the convolutions are
written as tap loops over shifted slices.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
import this file.  Pinned by tests/golden (make_golden.py imports the reference) through tests/test_oracle_golden.py.
"""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch

from . import separator_oracle as O
from .dualpath_oracle import gru_rnn, layer_norm, linear, lstm

SD = Dict[str, torch.Tensor]


def activation(x: torch.Tensor, kind: str, sd: SD, p: str) -> torch.Tensor:
    """lobe/activation.py: relu | prelu (one shared slope, key `weight`) | mish | sigmoid | tanh."""
    k = kind.lower()
    if k == "relu":
        return torch.relu(x)
    if k == "prelu":
        w = sd[p + "weight"].to(x.dtype)
        return torch.where(x >= 0, x, w * x)
    if k == "mish":
        return x * torch.tanh(torch.nn.functional.softplus(x))
    if k == "sigmoid":
        return torch.sigmoid(x)
    if k == "tanh":
        return torch.tanh(x)
    raise NameError("Could not interpret activation identifier")


def norm2d(x: torch.Tensor, kind: str, sd: SD, p: str) -> torch.Tensor:
    """bN2d in eval mode (running statistics) or gLN over [ch, F, T] (lobe/norm.py:20-34,95)."""
    if kind == "bN2d":
        sh = (1, -1, 1, 1)
        scale = sd[p + "weight"].to(x.dtype) / torch.sqrt(sd[p + "running_var"].to(x.dtype) + 1e-5)
        return (x - sd[p + "running_mean"].to(x.dtype).reshape(sh)) * scale.reshape(sh) + sd[p + "bias"].to(x.dtype).reshape(sh)
    if kind == "gLN":
        return O.glob_ln(x, sd[p + "gamma"], sd[p + "beta"])
    raise NotImplementedError(kind)


def conv2d(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], stride: Sequence[int], dilation: Sequence[int],
           pad_f: Sequence[int], pad_t: Sequence[int]) -> torch.Tensor:
    """nn.ZeroPad2d((pad_t[0], pad_t[1], pad_f[0], pad_f[1])) + nn.Conv2d(kernel (kf, kt), stride, dilation) on
    [N, Cin, F, T] (unet.py:112-128)."""
    n, cin, f, t = x.shape
    cout, _, kf, kt = w.shape
    xp = torch.zeros(n, cin, f + pad_f[0] + pad_f[1], t + pad_t[0] + pad_t[1], dtype=x.dtype)
    xp[:, :, pad_f[0]:pad_f[0] + f, pad_t[0]:pad_t[0] + t] = x
    fo = (xp.shape[2] - dilation[0] * (kf - 1) - 1) // stride[0] + 1
    to = (xp.shape[3] - dilation[1] * (kt - 1) - 1) // stride[1] + 1
    y = torch.zeros(n, cout, fo, to, dtype=x.dtype)
    for jf in range(kf):
        for jt in range(kt):
            sl = xp[:, :, jf * dilation[0]: jf * dilation[0] + (fo - 1) * stride[0] + 1: stride[0],
                    jt * dilation[1]: jt * dilation[1] + (to - 1) * stride[1] + 1: stride[1]]
            y = y + torch.einsum("oc,ncft->noft", w[:, :, jf, jt].to(x.dtype), sl)
    if b is not None:
        y = y + b.to(x.dtype).reshape(1, -1, 1, 1)
    return y


def conv_transpose2d(x: torch.Tensor, w: torch.Tensor, b: Optional[torch.Tensor], stride: Sequence[int],
                     dilation: Sequence[int], padding: Sequence[int], output_padding: Sequence[int]) -> torch.Tensor:
    """nn.ConvTranspose2d with weight [Cin, Cout, kf, kt] (unet.py:139-165): scatter every input element to
    out[(f*sf - pf + jf*df), (t*st - pt + jt*dt)]."""
    n, cin, f, t = x.shape
    _, cout, kf, kt = w.shape
    fo = (f - 1) * stride[0] - 2 * padding[0] + dilation[0] * (kf - 1) + output_padding[0] + 1
    to = (t - 1) * stride[1] - 2 * padding[1] + dilation[1] * (kt - 1) + output_padding[1] + 1
    full = torch.zeros(n, cout, fo + 2 * padding[0], to + 2 * padding[1], dtype=x.dtype)
    for jf in range(kf):
        for jt in range(kt):
            contrib = torch.einsum("co,ncft->noft", w[:, :, jf, jt].to(x.dtype), x)
            f0, t0 = jf * dilation[0], jt * dilation[1]
            # positions f*sf + f0 (before removing the padding); clip to the padded canvas
            fi = torch.arange(f) * stride[0] + f0
            ti = torch.arange(t) * stride[1] + t0
            fm, tm = fi < full.shape[2], ti < full.shape[3]
            full[:, :, fi[fm][:, None], ti[tm][None, :]] += contrib[:, :, fm][:, :, :, tm]
    y = full[:, :, padding[0]:padding[0] + fo, padding[1]:padding[1] + to]
    if b is not None:
        y = y + b.to(x.dtype).reshape(1, -1, 1, 1)
    return y


def unet_geometry(args: dict):
    ch = list(args["channels"])
    if args["input_type"].lower() == "ri":
        ch[0] = ch[0] * 2
    elif args["input_type"].lower() != "real":
        raise TypeError("Input feature type should be RI-concate, RI-stack or Real")
    kernel = list(zip(args["kernel_f"], args["kernel_t"]))
    dilation = list(zip(args["dilation_f"], args["dilation_t"]))
    stride = list(zip(args["stride_f"], args["stride_t"]))
    return ch, kernel, dilation, stride


def unet_down(x: torch.Tensor, sd: SD, p: str, args: dict) -> List[torch.Tensor]:
    """RI split + the CNN-down stack (unet.py:231-246) -> skip list (input first)."""
    ch, kernel, dilation, stride = unet_geometry(args)
    if args["input_type"].lower() == "ri":
        re, im = torch.chunk(x, 2, dim=-2)
        x = torch.stack([re, im], dim=1)
    elif x.dim() == 3:
        x = x.unsqueeze(1)
    skip = [x]
    for i in range(len(kernel)):
        kf, kt = kernel[i]
        lp = f"{p}cnn_down.{i}."
        x = conv2d(x, sd[lp + "1.weight"], sd[lp + "1.bias"], stride[i], dilation[i], (kf // 2, kf // 2),
                   (kt - args["delay"][i] - 1, args["delay"][i]))
        x = norm2d(x, args["norm_type"], sd, lp + "2.")
        x = activation(x, args["activation_type"], sd, lp + "3.")
        skip.append(x)
    return skip


def unet_up(x: torch.Tensor, skip: List[torch.Tensor], sd: SD, p: str, args: dict) -> torch.Tensor:
    """CNN-up stack with skip connections, time trimming and RI re-assembly (unet.py:248-283, 514-537)."""
    ch, kernel, dilation, stride = unet_geometry(args)
    n_cnn = len(kernel)
    tk = args["transpose_t_size"]
    for j, i in enumerate(reversed(range(n_cnn))):
        lp = f"{p}cnn_up.{j}."
        if args.get("skip_conv", False):
            sp = f"{p}skip_cnn.{j}."
            s = skip[-j - 1]
            s = torch.einsum("oc,ncft->noft", sd[sp + "0.weight"][:, :, 0, 0].to(x.dtype), s) + \
                sd[sp + "0.bias"].to(x.dtype).reshape(1, -1, 1, 1)
            x = x + activation(s, args["activation_type"], sd, sp + "1.")
        else:
            x = torch.cat([x, skip[-j - 1]], dim=1)
        sf, _ = stride[i]
        k = kernel[i][0]
        pd = k // 2
        op = sf - k + 2 * pd
        x = conv_transpose2d(x, sd[lp + "0.weight"], sd[lp + "0.bias"], stride[i], dilation[i], (pd, 0), (op, 0))
        if i != 0:
            x = norm2d(x, args["norm_type"], sd, lp + "1.")
            x = activation(x, args["activation_type"], sd, lp + "2.")
        if tk != 1:
            x = x[..., (tk - 1):] if args.get("transpose_delay", False) else x[..., :-(tk - 1)]
    mo = args.get("multi_output", 1)
    if mo != 1:
        b, c, fd, td = x.shape
        x = x.reshape(b, mo, -1, fd, td)
        return torch.cat([x[:, :, 0], x[:, :, 1]], dim=2) if args["input_type"].lower() == "ri" else x.squeeze(2)
    if args["input_type"].lower() == "ri":
        return torch.cat([x[:, 0], x[:, 1]], dim=1)
    return x.squeeze(1)


def unet(x: torch.Tensor, sd: SD, p: str, args: dict) -> torch.Tensor:
    """Unet.forward (unet.py:221-283)."""
    skip = unet_down(x, sd, p, args)
    return unet_up(skip[-1], skip, sd, p, args)


def unet_tcn(x: torch.Tensor, sd: SD, p: str, args: dict, dvec: Optional[torch.Tensor] = None) -> torch.Tensor:
    """UnetTcn.forward (unet.py:454-523)."""
    if args.get("embed_norm", False) and dvec is not None:
        dvec = dvec / dvec.norm(p=2, dim=1, keepdim=True).clamp_min(1e-12)
    skip = unet_down(x, sd, p, args)
    y = skip[-1]
    n, c, f, t = y.shape
    y = y.reshape(n, c * f, t)
    layer = args["tcn_layer"].lower()
    for r in range(args["repeat_tcn"]):
        for i in range(args["per_tcn_stack"]):
            bp = f"{p}tcn_list.{r}.{i}."
            e = dvec if args["tcn_with_embed"][i] else None
            d = args["tcn_dilated_basic"] ** i
            if layer == "normal":
                y = O.tcn_block(y, sd, bp, args["tcn_kernel"], d, args["causal"], args["tcn_norm"], args["dconv_norm"], e)
            elif layer == "gated":
                y = O.gated_tcn_block(y, sd, bp, args["tcn_kernel"], d, args["causal"], args["tcn_norm"],
                                      args.get("tcn_use_film", False) and bool(args["tcn_with_embed"][i]), e)
            else:
                raise NameError
    return unet_up(y.reshape(n, c, f, t), skip, sd, p, args)


def single_rnn(x: torch.Tensor, sd: SD, p: str, bidirectional: bool, kind: str = "LSTM") -> torch.Tensor:
    """SingleRNN(kind) (lobe/rnn.py:9-55): x [B, C, L] -> LSTM / GRU / RNN over L -> Linear -> [B, C, L]."""
    if kind == "LSTM":
        y, _ = lstm(x.transpose(1, 2), sd, p + "rnn.", bidirectional)
    else:
        y = gru_rnn(x.transpose(1, 2), sd, p + "rnn.", bidirectional, kind)
    return linear(y, sd[p + "proj.weight"], sd[p + "proj.bias"]).transpose(1, 2)


def dprnn_block2d(x: torch.Tensor, sd: SD, p: str, intra_skip: bool = True, inter_skip: bool = True) -> torch.Tensor:
    """DPRNNblock2D.forward (dpcrn.py:34-81): bidirectional LSTM along frequency per frame, LayerNorm over channels,
    skip; unidirectional LSTM along time per frequency bin, LayerNorm, skip (each skip under its flag, dpcrn.py:62-63, 78-79)."""
    n, ch, c, t = x.shape
    y = x.transpose(1, -1).reshape(n * t, c, ch)                       # [N*T, C, CH]
    y = single_rnn(y.permute(0, 2, 1), sd, p + "intra_rnn.", True).permute(0, 2, 1)
    y = layer_norm(y, sd[p + "intra_norm.weight"], sd[p + "intra_norm.bias"])
    y = y.reshape(n, t, c, ch).transpose(1, -1)
    x = x + y if intra_skip else y
    y = x.permute(0, 2, 3, 1).reshape(n * c, t, ch)                    # [N*C, T, CH]
    y = single_rnn(y.permute(0, 2, 1), sd, p + "inter_rnn.", False).permute(0, 2, 1)
    y = layer_norm(y, sd[p + "inter_norm.weight"], sd[p + "inter_norm.bias"])
    y = y.permute(0, 2, 1).reshape(n, c, ch, t).permute(0, 2, 1, 3)
    return x + y if inter_skip else y


def dpcrn(x: torch.Tensor, sd: SD, p: str, args: dict) -> torch.Tensor:
    """DPCRN.forward (dpcrn.py:135-191), spectral_compress=False."""
    if args.get("spectral_compress", False):
        raise NotImplementedError("spectral_compress")
    skip = unet_down(x, sd, p, args)
    y = dprnn_block2d(skip[-1], sd, p + "dprnn_block1.")
    y = dprnn_block2d(y, sd, p + "dprnn_block2.")
    return unet_up(y, skip, sd, p, args)


# ---------------------------------------------------------------------------
# DPARN (dparn.py:12-247, lobe/attention.py:8-232)
# ---------------------------------------------------------------------------
def positional_table(length: int, d_model: int, dtype=torch.float32) -> torch.Tensor:
    """PositionalEncoding.pe[:length, 0] (lobe/attention.py:17-24): sin on even, cos on odd feature indices."""
    import math
    pos = torch.arange(length, dtype=torch.float64).unsqueeze(1)
    div = torch.exp(torch.arange(0, d_model, 2, dtype=torch.float64) * (-math.log(10000.0) / d_model))
    pe = torch.zeros(length, d_model, dtype=torch.float64)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.to(dtype)


def multihead_attention(x: torch.Tensor, w_in: torch.Tensor, w_out: torch.Tensor, heads: int, causal: bool = False):
    """nn.MultiheadAttention(E, heads, bias=False, batch_first=True) self-attention on x [B, L, E]
    (lobe/attention.py:38-112): q,k,v = x W_in^T split in three, per head softmax(q k^T / sqrt(dh)) v, then W_out."""
    b, l, e = x.shape
    dh = e // heads
    qkv = linear(x, w_in)
    q, k, v = [t.reshape(b, l, heads, dh).transpose(1, 2) for t in qkv.split(e, dim=-1)]   # [B, h, L, dh]
    s = torch.matmul(q, k.transpose(-1, -2)) / (dh ** 0.5)
    if causal:
        s = s + torch.triu(torch.full((l, l), float("-inf"), dtype=x.dtype), diagonal=1)
    p = torch.softmax(s, dim=-1)
    o = torch.matmul(p, v).transpose(1, 2).reshape(b, l, e)
    return linear(o, w_out)


def mha_self_atten_layer(x: torch.Tensor, sd: SD, p: str, heads: int, position_encoding: bool,
                         causal: bool = False, improved: bool = False, bidirectional: bool = False) -> torch.Tensor:
    """MhaSelfAttenLayer.forward (lobe/attention.py:180-232): x [B, C, L] -> [B, C, L].  improved: the first
    feed-forward Linear is an LSTM over the sequence (:170-183, :221-222)."""
    y = x.transpose(1, 2)
    src = y
    if position_encoding:
        # `pe` is a persistent buffer of the reference (it travels in checkpoints); its values are positional_table()
        y = y + sd[p + "pos.pe"][:y.shape[1], 0].to(y.dtype).unsqueeze(0)
    a = multihead_attention(y, sd[p + "self_atten.atten.in_proj_weight"], sd[p + "self_atten.atten.out_proj.weight"],
                            heads, causal)
    y = layer_norm(src + a, sd[p + "norm1.weight"], sd[p + "norm1.bias"])
    if improved:
        from . import dualpath_oracle as DP
        r, _ = DP.lstm(y, sd, p + "recurrent.", bidirectional)
        ff = linear(torch.relu(r), sd[p + "feedforward.2.weight"], sd[p + "feedforward.2.bias"])
    else:
        ff = linear(torch.relu(linear(y, sd[p + "feedforward.0.weight"], sd[p + "feedforward.0.bias"])),
                    sd[p + "feedforward.3.weight"], sd[p + "feedforward.3.bias"])
    return layer_norm(y + ff, sd[p + "norm2.weight"], sd[p + "norm2.bias"]).transpose(1, 2)


def dparn_block2d(x: torch.Tensor, sd: SD, p: str, heads: int, intra_skip: bool = True, inter_skip: bool = True) -> torch.Tensor:
    """DPARNblock2D.forward (dparn.py:55-108): two self-attention layers along frequency per frame, Linear,
    LayerNorm, skip; then the unidirectional LSTM along time of the DPCRN block (each skip under its flag)."""
    n, ch, c, t = x.shape
    y = x.transpose(1, -1).reshape(n * t, c, ch).permute(0, 2, 1)       # [N*T, CH, C]
    y = mha_self_atten_layer(y, sd, p + "intra_atten1.", heads, True)
    y = mha_self_atten_layer(y, sd, p + "intra_atten2.", heads, False)
    y = linear(y.permute(0, 2, 1), sd[p + "intra_fc.weight"], sd[p + "intra_fc.bias"])
    y = layer_norm(y, sd[p + "intra_norm.weight"], sd[p + "intra_norm.bias"])
    y = y.reshape(n, t, c, ch).transpose(1, -1)
    x = x + y if intra_skip else y
    y = x.permute(0, 2, 3, 1).reshape(n * c, t, ch)
    y = single_rnn(y.permute(0, 2, 1), sd, p + "inter_rnn.", False).permute(0, 2, 1)
    y = layer_norm(y, sd[p + "inter_norm.weight"], sd[p + "inter_norm.bias"])
    y = y.permute(0, 2, 1).reshape(n, c, ch, t).permute(0, 2, 1, 3)
    return x + y if inter_skip else y


def dparn(x: torch.Tensor, sd: SD, p: str, args: dict) -> torch.Tensor:
    """DPARN.forward (dparn.py:170-226), spectral_compress=False."""
    if args.get("spectral_compress", False):
        raise NotImplementedError("spectral_compress")
    skip = unet_down(x, sd, p, args)
    y = dparn_block2d(skip[-1], sd, p + "dprnn_block1.", args["nhead"])
    y = dparn_block2d(y, sd, p + "dprnn_block2.", args["nhead"])
    return unet_up(y, skip, sd, p, args)
