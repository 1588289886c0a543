"""DPARN on the HIP path (mirror of puresound/nnet/dparn.py:12-247): DPCRN with the intra-frame LSTM replaced by two
self-attention layers along frequency (lobe/attention.py), then Linear + LayerNorm + skip; the inter pass is the
unidirectional LSTM along time of DPCRN.  Same rows as dpcrn.py: [N][CH] feature rows over F*ld frames."""
from typing import Dict, Tuple

import torch
import torch.nn as nn

from .. import hip
from ..ops import op_module, same_shape
from ._plans import PlanCache, layernorm_plan, linear_plan, lstm_path, lstm_plan
from .lobe.attention import MhaSelfAttenLayer
from .lobe.rnn import SingleRNN
from .unet import Unet, _unet_shape


class DPARNblock2D(PlanCache, nn.Module):
    """dparn.py:12-108."""

    def __init__(self, input_size: int, hidden_size: int, nhead: int, dropout: float = 0.0) -> None:
        super().__init__()
        self.intra_atten1 = MhaSelfAttenLayer(input_size, hidden_size, nhead=nhead, dropout=dropout, improved=False,
                                              bidirectional=False, position_encoding=True)
        self.intra_atten2 = MhaSelfAttenLayer(input_size, hidden_size, nhead=nhead, dropout=dropout, improved=False,
                                              bidirectional=False, position_encoding=False)
        self.intra_fc = nn.Linear(input_size, input_size)
        self.intra_norm = nn.LayerNorm(input_size)
        self.inter_rnn = SingleRNN("LSTM", input_size, hidden_size, bidirectional=False, dropout=dropout)
        self.inter_norm = nn.LayerNorm(input_size)

    def _build(self, device):
        if self.training and self.inter_rnn.drop.p > 0:
            raise RuntimeError("DPARNblock2D: dropout is active; the HIP path is inference only -- call .eval()")
        return dict(fc=linear_plan(self.intra_fc, device), fc_norm=layernorm_plan(self.intra_norm, device),
                    inter=(lstm_plan(self.inter_rnn.rnn, device, self.gemm_precision), linear_plan(self.inter_rnn.proj, device),
                           layernorm_plan(self.inter_norm, device)))

    def forward_padded(self, x: torch.Tensor, t: int, amax=None, intra_skip: bool = True, inter_skip: bool = True) -> torch.Tensor:
        """[N, CH, F, ld] -> [N, CH, F, ld].  amax: lstm_path's one-element list (fp16x2 arithmetic: the maxima of |x| travel
        from block to block)."""
        p = self._plan_get(x.device, self._build)
        n, ch, f, ld = x.shape
        y = x.view(n, ch, f * ld)
        frames = (f - 1) * ld + t
        amax = [None] if amax is None else amax
        a, bound = self.intra_atten1.forward_padded(y, frames, t, 1, f, ld, x_amax=amax[0], want_bound=True)
        a, bound = self.intra_atten2.forward_padded(a, frames, t, 1, f, ld, want_bound=True, x_bound=bound or 0.0)
        fn = p["fc_norm"]
        amax[0] = None
        if (self.gemm_precision == "fp16x2" and bound is not None and ch == 128 and hip.conv1x1_f16x2_ln_ok(n, ch, 128, frames)):
            # Linear + LayerNorm + skip as one launch (the LayerNorm epilogue of the fp16x2 GEMM); its maxima feed the recurrence
            if "f16x2_ln" not in p["fc"]:
                w256 = torch.zeros(256, ch, dtype=torch.float32, device=x.device)
                w256[:ch] = p["fc"]["w_rows"]
                p["fc"]["f16x2_ln"] = hip.pack_wt_f16x2(w256)
            wf, we = p["fc"]["f16x2_ln"]
            y, amax[0] = hip.conv1x1_f16x2_ln(a, frames, wf, we, ch, p["fc"]["bias"], fn["gamma"], fn["beta"], fn["eps"],
                                              y if intra_skip else None, x_bound=bound, want_amax=True)
        else:
            y, _ = hip.proj_layernorm(a, frames, p["fc"]["wt"], p["fc"]["bias"], ch, fn["gamma"], fn["beta"], fn["eps"],
                                      y if intra_skip else None)
        y, _ = lstm_path(y, frames, *p["inter"], q=f, q_stride=ld, steps=t, step_stride=1, amax=amax, skip=inter_skip)
        return y.view(n, ch, f, ld)

    def forward(self, x: torch.Tensor, intra_skip: bool = True, inter_skip: bool = True) -> torch.Tensor:
        hip.require_device(x, "DPARNblock2D.forward")
        n, ch, f, t = x.shape
        y = self.forward_padded(hip.pad_rows(x.reshape(n, ch * f, t)).view(n, ch, f, -1), t, None, intra_skip, inter_skip)
        return hip.unpad_rows(y.reshape(n, ch * f, -1), t).view(n, ch, f, t)


@op_module("dparn_fwd", same_shape)
class DPARN(Unet):
    """dparn.py:110-247; constructor order as the reference (dparn.py:111-131)."""

    def __init__(self, input_type: str = "RI", input_dim: int = 512, activation_type: str = "PReLU",
                 norm_type: str = "bN2d", dropout: float = 0.05, channels: Tuple = (1, 32, 32, 32, 64, 128),
                 transpose_t_size: int = 2, transpose_delay: bool = False, skip_conv: bool = False,
                 kernel_t: Tuple = (2, 2, 2, 2, 2), stride_t: Tuple = (1, 1, 1, 1, 1),
                 dilation_t: Tuple = (1, 1, 1, 1, 1), kernel_f: Tuple = (5, 3, 3, 3, 3),
                 stride_f: Tuple = (2, 2, 1, 1, 1), dilation_f: Tuple = (1, 1, 1, 1, 1), delay: Tuple = (0, 0, 0, 0, 0),
                 rnn_hidden: int = 128, nhead: int = 1, spectral_compress: bool = False, _multi_output: int = 1):
        super().__init__(input_type, input_dim, activation_type, norm_type, dropout, channels, transpose_t_size,
                         skip_conv, kernel_t, stride_t, dilation_t, kernel_f, stride_f, dilation_f, delay, _multi_output)
        self.transpose_delay = transpose_delay
        self.rnn_hidden = rnn_hidden
        self.spectral_compress = spectral_compress
        self.dprnn_block1 = DPARNblock2D(input_size=channels[-1], hidden_size=rnn_hidden, nhead=nhead, dropout=dropout)
        self.dprnn_block2 = DPARNblock2D(input_size=channels[-1], hidden_size=rnn_hidden, nhead=nhead, dropout=dropout)

    def forward_padded4(self, x4: torch.Tensor, t: int, dvec=None) -> torch.Tensor:
        if self.spectral_compress:
            raise NotImplementedError("DPARN on HIP: spectral_compress (it returns a complex tensor in the reference)")
        p = self._plan_get(x4.device, self._build_unet)
        skip = self._down(x4, t, p)
        amax = [None]
        y = self.dprnn_block1.forward_padded(skip[-1], t, amax)
        y = self.dprnn_block2.forward_padded(y, t, amax)
        return self._up(y, skip, t, p, self.transpose_delay)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [N, C, T] -> [N, C, T] (dparn.py:170-226)."""
        hip.require_device(x, "DPARN.forward")
        x4, t = self._split_in(x)
        return self._merge_out(self.forward_padded4(x4, t), t)

    @property
    def get_args(self) -> Dict:
        a = dict(Unet.get_args.fget(self))
        a.pop("multi_output")
        a.update(transpose_delay=self.transpose_delay, rnn_hidden=self.rnn_hidden)
        return a


@op_module("dparn_mout_fwd", _unet_shape)
class DPARN_Mout(DPARN):
    """dparn.py:249-401: DPARN whose last transposed convolution carries `multi_output` masks; the output is
    [N, multi_output, C, T] (the masks are consecutive channel groups of the decoder's output rows, so the reshape of
    dparn.py:351-358 is a view of them).  Constructor order as the reference (dparn.py:250-272)."""

    def __init__(self, input_type: str = "RI", input_dim: int = 512, activation_type: str = "PReLU",
                 norm_type: str = "bN2d", dropout: float = 0.05, channels: Tuple = (1, 32, 32, 32, 64, 128),
                 transpose_t_size: int = 2, transpose_delay: bool = False, skip_conv: bool = False,
                 kernel_t: Tuple = (2, 2, 2, 2, 2), stride_t: Tuple = (1, 1, 1, 1, 1),
                 dilation_t: Tuple = (1, 1, 1, 1, 1), kernel_f: Tuple = (5, 3, 3, 3, 3),
                 stride_f: Tuple = (2, 2, 1, 1, 1), dilation_f: Tuple = (1, 1, 1, 1, 1), delay: Tuple = (0, 0, 0, 0, 0),
                 multi_output: int = 2, rnn_hidden: int = 128, nhead: int = 1, spectral_compress: bool = False):
        super().__init__(input_type, input_dim, activation_type, norm_type, dropout, channels, transpose_t_size,
                         transpose_delay, skip_conv, kernel_t, stride_t, dilation_t, kernel_f, stride_f, dilation_f,
                         delay, rnn_hidden, nhead, spectral_compress, _multi_output=multi_output)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [N, C, T] -> [N, multi_output, C, T] (dparn.py:308-376)."""
        hip.require_device(x, "DPARN_Mout.forward")
        x4, t = self._split_in(x)
        return self._merge_out(self.forward_padded4(x4, t), t)

    @property
    def get_args(self) -> Dict:
        """dparn.py:378-401 (as there: nhead is not listed)."""
        a = dict(Unet.get_args.fget(self))
        a.update(transpose_delay=self.transpose_delay, rnn_hidden=self.rnn_hidden,
                 spectral_compress=self.spectral_compress)
        return a
