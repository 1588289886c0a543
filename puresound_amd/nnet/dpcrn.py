"""DPCRN on the HIP path (mirror of puresound/nnet/dpcrn.py:11-213): the U-Net convolutions of unet.py around two
2-D dual-path blocks.  On the [N, CH, F, ld] rows the bottleneck is [N][CH] feature rows over F*ld frames, so
DPRNNblock2D is the same three kernels as the 1-D DPRNN: input-projection GEMM, ps_lstm_f32 (intra: one sequence per
frame walking the F rows, step_stride = ld; inter: one sequence per frequency row walking its T frames) and
projection + LayerNorm + residual.
"""
from typing import Dict, Tuple

import torch
import torch.nn as nn

from .. import hip
from ..ops import op_module, same_shape
from ._plans import PlanCache, layernorm_plan, linear_plan, lstm_path, lstm_plan
from .lobe.rnn import SingleRNN
from .unet import Unet


class DPRNNblock2D(PlanCache, nn.Module):
    """dpcrn.py:11-81."""

    def __init__(self, input_size: int, hidden_size: int, dropout: float = 0.0) -> None:
        super().__init__()
        self.intra_rnn = SingleRNN("LSTM", input_size, hidden_size, bidirectional=True, dropout=dropout)
        self.intra_norm = nn.LayerNorm(input_size)
        self.inter_rnn = SingleRNN("LSTM", input_size, hidden_size, bidirectional=False, dropout=dropout)
        self.inter_norm = nn.LayerNorm(input_size)

    def _build(self, device):
        if self.training and (self.intra_rnn.drop.p > 0 or self.inter_rnn.drop.p > 0):
            raise RuntimeError("DPRNNblock2D: dropout is active; the HIP path is inference only -- call .eval()")
        return dict(intra=(lstm_plan(self.intra_rnn.rnn, device, self.gemm_precision), linear_plan(self.intra_rnn.proj, device),
                           layernorm_plan(self.intra_norm, device)),
                    inter=(lstm_plan(self.inter_rnn.rnn, device, self.gemm_precision), linear_plan(self.inter_rnn.proj, device),
                           layernorm_plan(self.inter_norm, device)))

    def forward_padded(self, x: torch.Tensor, t: int, amax=None, intra_skip: bool = True, inter_skip: bool = True) -> torch.Tensor:
        """[N, CH, F, ld] -> [N, CH, F, ld].  amax: the one-element list of lstm_path (fp16x2 arithmetic: the maxima of |x|
        travel from recurrence to recurrence and from block to block instead of being measured before every GEMM)."""
        p = self._plan_get(x.device, self._build)
        n, ch, f, ld = x.shape
        y = x.view(n, ch, f * ld)
        frames = (f - 1) * ld + t                      # frames of the flattened (f, t) axis that hold data
        amax = [None] if amax is None else amax
        y, _ = lstm_path(y, frames, *p["intra"], q=t, q_stride=1, steps=f, step_stride=ld, amax=amax, skip=intra_skip)
        y, _ = lstm_path(y, frames, *p["inter"], q=f, q_stride=ld, steps=t, step_stride=1, amax=amax, skip=inter_skip)
        return y.view(n, ch, f, ld)

    def forward(self, x: torch.Tensor, intra_skip: bool = True, inter_skip: bool = True) -> torch.Tensor:
        hip.require_device(x, "DPRNNblock2D.forward")
        n, ch, f, t = x.shape
        y = self.forward_padded(hip.pad_rows(x.reshape(n, ch * f, t)).view(n, ch, f, -1), t, None, intra_skip, inter_skip)
        return hip.unpad_rows(y.reshape(n, ch * f, -1), t).view(n, ch, f, t)


@op_module("dpcrn_fwd", same_shape)
class DPCRN(Unet):
    """dpcrn.py:84-213; constructor order as the reference (dpcrn.py:85-104)."""

    def __init__(self, input_type: str = "RI", input_dim: int = 512, activation_type: str = "PReLU",
                 norm_type: str = "bN2d", dropout: float = 0.05, channels: Tuple = (1, 32, 32, 32, 64, 128),
                 transpose_t_size: int = 2, transpose_delay: bool = False, skip_conv: bool = False,
                 kernel_t: Tuple = (2, 2, 2, 2, 2), stride_t: Tuple = (1, 1, 1, 1, 1),
                 dilation_t: Tuple = (1, 1, 1, 1, 1), kernel_f: Tuple = (5, 3, 3, 3, 3),
                 stride_f: Tuple = (2, 2, 1, 1, 1), dilation_f: Tuple = (1, 1, 1, 1, 1), delay: Tuple = (0, 0, 0, 0, 0),
                 rnn_hidden: int = 128, spectral_compress: bool = False):
        super().__init__(input_type, input_dim, activation_type, norm_type, dropout, channels, transpose_t_size,
                         skip_conv, kernel_t, stride_t, dilation_t, kernel_f, stride_f, dilation_f, delay)
        self.transpose_delay = transpose_delay
        self.rnn_hidden = rnn_hidden
        self.spectral_compress = spectral_compress
        self.dprnn_block1 = DPRNNblock2D(input_size=channels[-1], hidden_size=rnn_hidden, dropout=dropout)
        self.dprnn_block2 = DPRNNblock2D(input_size=channels[-1], hidden_size=rnn_hidden, dropout=dropout)

    def forward_padded4(self, x4: torch.Tensor, t: int, dvec=None) -> torch.Tensor:
        """[N, CH0, F, ld] -> [N, CH0, F, ld]."""
        if self.spectral_compress:
            raise NotImplementedError("DPCRN on HIP: spectral_compress (it returns a complex tensor in the reference)")
        p = self._plan_get(x4.device, self._build_unet)
        skip = self._down(x4, t, p)
        amax = [None]
        y = self.dprnn_block1.forward_padded(skip[-1], t, amax)
        y = self.dprnn_block2.forward_padded(y, t, amax)
        return self._up(y, skip, t, p, self.transpose_delay)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [N, C, T] -> [N, C, T] (dpcrn.py:135-191)."""
        hip.require_device(x, "DPCRN.forward")
        x4, t = self._split_in(x)
        return self._merge_out(self.forward_padded4(x4, t), t)

    @property
    def get_args(self) -> Dict:
        a = dict(Unet.get_args.fget(self))
        a.pop("multi_output")
        a.update(transpose_delay=self.transpose_delay, rnn_hidden=self.rnn_hidden)
        return a
