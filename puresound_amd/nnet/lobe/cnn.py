"""Depthwise-separable conv lobe (mirror of puresound/nnet/lobe/cnn.py:9-106).

Holds the parameters under the reference's keys (depthwise.{0,1,2}, pointwise.{0,1,2}, optional
in_conv / skip_conv).  The arithmetic lives in ps_dwconv_f32 and ps_conv1x1_f32 and is driven by the
enclosing TCN block; the hid_channels / skip variants are not used by any Conv-TasNet preset.
"""
from typing import Optional

import torch
import torch.nn as nn

from .norm import get_norm


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
        raise NotImplementedError(
            "DepthwiseSeparableConv1d runs fused inside TCN.forward on the HIP path; call the TCN block")
