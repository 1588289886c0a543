"""Depthwise-separable conv lobe (mirror of puresound/nnet/lobe/cnn.py:9-106).

Holds the parameters under the reference's keys (depthwise.{0,1,2}, pointwise.{0,1,2}, optional
in_conv / skip_conv).  Inside a TCN block the arithmetic is driven by the block's fused driver
(ps_conv_tasnet_f32); called on its own -- with the hid_channels transform and the skip connection, which no
Conv-TasNet preset uses -- `forward` runs the same kernels stage by stage: every norm + PReLU is the consumer-side
prologue of the next kernel, the last one is applied by ps_norm_activation_f32 / ps_chan_layernorm_f32.
"""
from typing import Optional

import torch
import torch.nn as nn

from ... import hip
from ..._abi import PS_NORM_GLOBAL
from .norm import ChanLN, get_norm, norm_plan


class DepthwiseSeparableConv1d(nn.Module):
    def __init__(self, in_channels: int, out_channels: int, hid_channels: Optional[int] = None,
                 norm_cls: str = "gGN", kernel: int = 3, stride: int = 1, dilation: int = 1,
                 skip: bool = False, causal: bool = False) -> None:
        super().__init__()
        self.skip = skip
        self.transform = False
        self.causal = causal
        if self.causal:
            # same conflict check as cnn.py:40-44
            assert norm_cls not in ["gLN", "gGN"], \
                "Conflict setting between normalization layer and causal operation."
        self.norm_name = norm_cls
        norm = get_norm(norm_cls)
        if hid_channels is not None:
            self.transform = True
            self.in_conv = nn.Sequential(nn.Conv1d(in_channels, hid_channels, 1), norm(hid_channels), nn.PReLU())
        self.hid_channels = hid_channels if hid_channels is not None else in_channels
        self.kernel, self.stride, self.dilation = kernel, stride, dilation
        self.padding = (kernel - 1) * dilation if self.causal else ((kernel - 1) // 2) * dilation
        self.depthwise = nn.Sequential(
            nn.Conv1d(self.hid_channels, self.hid_channels, kernel_size=kernel, stride=stride, dilation=dilation,
                      padding=self.padding, groups=self.hid_channels),
            norm(self.hid_channels), nn.PReLU())
        self.pointwise = nn.Sequential(nn.Conv1d(self.hid_channels, out_channels, kernel_size=1, stride=1),
                                       norm(out_channels), nn.PReLU())
        if self.skip:
            self.skip_conv = nn.Conv1d(in_channels, out_channels, 1)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [N, C, T] -> [N, out_channels, T] (cnn.py:84-106): [in_conv] -> depthwise -> pointwise (each conv + norm +
        PReLU) [+ skip_conv(x)].  Causal: the reference pads both sides and cuts the tail, i.e. left padding only."""
        hip.require_device(x, "DepthwiseSeparableConv1d.forward")
        if self.stride != 1 and self.skip:
            # the reference adds skip_conv(x) of length T to a result of length ~T / stride: torch's broadcasting error
            raise RuntimeError("The size of tensor a must match the size of tensor b at non-singleton dimension 2 "
                               "(DepthwiseSeparableConv1d: skip=True needs stride=1)")
        if not self.causal and self.kernel % 2 == 0:
            raise RuntimeError("DepthwiseSeparableConv1d: an even kernel with symmetric padding changes the length")
        with torch.no_grad():
            n, _, t = x.shape
            f32 = dict(dtype=torch.float32, device=x.device)
            xp = hip.pad_rows(x.float())
            ldt = xp.shape[-1]

            def parts(seq):
                """conv weight [M, K], bias, (kind, gamma, beta) of the norm, PReLU slope of one conv + norm + PReLU"""
                conv, norm, act = seq
                if act.weight.numel() != 1:
                    raise NotImplementedError("PReLU with per-channel slopes is not on the HIP path")
                if isinstance(norm, ChanLN):
                    nk = ("cln", norm.gamma.detach().to(**f32).contiguous(), norm.beta.detach().to(**f32).contiguous())
                else:
                    kind, g, b = norm_plan(norm)
                    nk = (kind, g.to(**f32).contiguous(), b.to(**f32).contiguous())
                return conv, nk, act.weight.detach().to(**f32).contiguous()

            keep = []  # a prologue holds raw pointers: its tensors stay alive until the last launch is enqueued

            def settle(y, stats, nk, slope, rows):
                """(tensor, prologue) the next kernel consumes for the norm + PReLU that follow `y`"""
                kind, g, b = nk
                keep.extend((y, stats, g, b, slope))
                if kind == "cln":
                    return hip.chan_layernorm(y, t, g, b, 1e-8, slope=slope), None
                return y, hip.make_prologue(kind, True, stats, rows * t, 1e-8, g, b, slope)

            a, pro = xp, None
            h = self.hid_channels
            if self.transform:
                conv, nk, slope = parts(self.in_conv)
                y, st = hip.conv1x1(xp, t, hip.pack_wt(conv.weight.detach().to(**f32)), h, None,
                                    conv.bias.detach().to(**f32), want_stats=nk[0] == PS_NORM_GLOBAL,
                                    out=torch.empty(n, h, ldt, **f32))
                a, pro = settle(y, st, nk, slope, h)
            conv, nk, slope = parts(self.depthwise)
            y, st = hip.dwconv(a, t, conv.weight.detach().to(**f32).contiguous(), conv.bias.detach().to(**f32).contiguous(),
                               self.dilation, self.padding, pro, nk[0] == PS_NORM_GLOBAL and self.stride == 1)
            if self.stride != 1:
                # Conv1d(stride=s) is every s-th frame of the stride-1 result (cnn.py:62-71): the stride-1 kernel, a strided
                # copy (no recipe takes this path), the norm's statistics over the frames that remain
                lf = t + 2 * self.padding - self.dilation * (self.kernel - 1)      # stride-1 output length
                keep_t = (lf + self.stride - 1) // self.stride
                if self.causal:
                    # both-sided padding, every s-th frame, the last `padding` frames cut at the end (cnn.py:100-101).  The
                    # causal norms are frame-local (gLN / gGN are refused by the constructor), so cutting here is the same;
                    # every frame that is kept lies inside the T frames the left-padded stride-1 kernel produces
                    keep_t -= self.padding
                    if keep_t <= 0:
                        raise RuntimeError(f"DepthwiseSeparableConv1d(causal, stride={self.stride}): {t} frames leave "
                                           f"none behind the cut of {self.padding} (the reference returns an empty tensor)")
                    lf = t
                y = hip.pad_rows(y[..., :lf][..., ::self.stride][..., :keep_t].contiguous())
                t = keep_t
                ldt = y.shape[-1]
                st = hip.row_stats(y, t) if nk[0] == PS_NORM_GLOBAL else None
            a, pro = settle(y, st, nk, slope, h)
            conv, nk, slope = parts(self.pointwise)
            m = conv.out_channels
            y, st = hip.conv1x1(a, t, hip.pack_wt(conv.weight.detach().to(**f32)), m, pro, conv.bias.detach().to(**f32),
                                want_stats=nk[0] == PS_NORM_GLOBAL, out=torch.empty(n, m, ldt, **f32))
            if nk[0] == "cln":
                y = hip.chan_layernorm(y, t, nk[1], nk[2], 1e-8, slope=slope)
            else:
                keep.extend((st, nk[1], nk[2], slope))
                last = hip.make_prologue(nk[0], False, st, m * t, 1e-8, nk[1], nk[2], None)
                hip.norm_activation_(y.view(n, m, 1, ldt), t, last, 0.0, 0.0, "prelu", slope)
            if self.skip:
                sc = self.skip_conv
                y, _ = hip.conv1x1(xp, t, hip.pack_wt(sc.weight.detach().to(**f32)), m, None, sc.bias.detach().to(**f32),
                                   res=y, out=torch.empty(n, m, ldt, **f32))
            return hip.unpad_rows(y, t)
