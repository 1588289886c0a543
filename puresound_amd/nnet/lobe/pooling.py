"""Attentive statistics pooling on the HIP path (mirror of puresound/nnet/lobe/pooling.py:58-126).

tdnn (1x1 conv -> ReLU -> BatchNorm1d) and the attention conv run on ps_conv1x1_f32 -- the ReLU, the folded
eval BatchNorm and the tanh are the second GEMM's prologue -- and ps_attn_stats_pool_f32 does the softmax over
frames and the weighted mean / std (`lengths`: ragged batches); `return_weight=True` returns the softmax map itself
(ps_attn_weights_f32).
"""
import torch
import torch.nn as nn

from ... import hip
from ...ops import op_module
from ..._abi import PS_NORM_AFFINE


def _pool_shape(ctor, x, aux, params):
    return (x[0], 2 * x[1], 1)


@op_module("attn_stats_pool_fwd", _pool_shape, method="_pool")
class AttentiveStatisticsPooling(nn.Module):
    # arithmetic of the two 1x1 convs: "fp16x2" (two fp16 terms per operand, three MFMA products, fp32 accumulation -- the
    # TCN blocks' default; the first conv measures its input's range, the second one's input is a tanh) or "fp32"
    gemm_precision = "fp16x2"

    def __init__(self, channels, attention_channels=128):
        super().__init__()
        self.eps = 1e-12
        self.channels, self.attention_channels = channels, attention_channels
        self.tdnn = nn.Sequential(
            nn.Conv1d(in_channels=channels, out_channels=attention_channels, kernel_size=1, dilation=1),
            nn.ReLU(), nn.BatchNorm1d(attention_channels))
        self.tanh = nn.Tanh()
        self.conv = nn.Conv1d(in_channels=attention_channels, out_channels=channels, kernel_size=1)
        self._plan = None

    def __getstate__(self):
        state = dict(self.__dict__)
        state["_plan"] = None
        return state

    def _get_plan(self, device):
        sig = tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers())) + \
            (self.training, str(device), self.gemm_precision)
        if self._plan is None or self._plan["sig"] != sig:
            bn = self.tdnn[2]
            if bn.training:
                raise RuntimeError("AttentiveStatisticsPooling: BatchNorm1d must be in eval() mode on the HIP path")
            f32 = dict(dtype=torch.float32, device=device)
            scale = (bn.weight.detach() / torch.sqrt(bn.running_var.detach() + bn.eps)).to(**f32).contiguous()
            shift = (bn.bias.detach().to(**f32) - bn.running_mean.detach().to(**f32) * scale).contiguous()
            self._plan = dict(sig=sig,
                              w1=hip.pack_wt(self.tdnn[0].weight.detach().to(**f32)),
                              b1=self.tdnn[0].bias.detach().to(**f32).contiguous(),
                              w2=hip.pack_wt(self.conv.weight.detach().to(**f32)),
                              b2=self.conv.bias.detach().to(**f32).contiguous(), scale=scale, shift=shift)
            if self.gemm_precision == "fp16x2" and min(self.channels, self.attention_channels) >= 64:
                self._plan["w1_f16x2"] = hip.pack_wt_f16x2(self.tdnn[0].weight.detach().to(**f32))
                self._plan["w2_f16x2"] = hip.pack_wt_f16x2(self.conv.weight.detach().to(**f32))
        return self._plan

    def forward_padded(self, x_pad: torch.Tensor, t: int, lengths=None, return_weight: bool = False) -> torch.Tensor:
        """padded [N,C,ldt] (+ relative lengths [N]) -> [N,2C] (mean ; std), or the padded attention map [N,C,ldt]."""
        p = self._get_plan(x_pad.device)
        n, _, ldt = x_pad.shape
        pro = hip.make_prologue(PS_NORM_AFFINE, False, None, 0.0, 0.0, p["scale"], p["shift"], None,
                                pre_relu=True, post_tanh=True)
        h = torch.empty(n, self.attention_channels, ldt, device=x_pad.device)
        logits = torch.empty(n, self.channels, ldt, device=x_pad.device)
        if "w1_f16x2" in p:
            (w1, e1), (w2, e2) = p["w1_f16x2"], p["w2_f16x2"]
            hip.conv1x1_f16x2(x_pad, t, w1, e1, self.attention_channels, None, p["b1"], out=h,
                              x_amax=hip.absmax(x_pad, t))
            hip.conv1x1_f16x2(h, t, w2, e2, self.channels, pro, p["b2"], out=logits, x_bound=1.0)  # |tanh| <= 1
        else:
            hip.conv1x1(x_pad, t, p["w1"], self.attention_channels, None, p["b1"], out=h)
            hip.conv1x1(h, t, p["w2"], self.channels, pro, p["b2"], out=logits)
        if return_weight:
            return hip.attn_weights(logits, t, lengths)
        return hip.attn_stats_pool(logits, x_pad, t, self.eps, lengths)

    def _pool(self, x: torch.Tensor, lengths=None):
        hip.require_device(x, "AttentiveStatisticsPooling.forward")
        return self.forward_padded(hip.pad_rows(x), x.shape[-1], lengths).unsqueeze(2)

    def forward(self, x: torch.Tensor, lengths=None, return_weight: bool = False):
        """x [N,C,L] (+ relative lengths [N]) -> [N,2C,1] (pooling.py:87-126)."""
        if return_weight:  # the attention map itself [N,C,L] (pooling.py:112-113)
            hip.require_device(x, "AttentiveStatisticsPooling.forward")
            with torch.no_grad():
                t = x.shape[-1]
                return hip.unpad_rows(self.forward_padded(hip.pad_rows(x), t, lengths, True), t)
        return self._pool(x, lengths)
