"""U-Net maskers on the HIP path (mirror of puresound/nnet/unet.py:13-557: Unet, UnetTcn).

Layout: a 4-D activation [N, CH, F, T] lives as [N][CH*F] rows of ld frames ([N, CH, F, ld] tensors).  Every
Conv2d / ConvTranspose2d is ps_unfold2d_f32 (taps side by side, zero padding, stride, the decoder's channel concat
and its time trim) + one ps_conv1x1_f32 GEMM over the flattened (f, t) axis with the eval BatchNorm2d folded into the
weight, then ps_activation_f32.  The TCN bottleneck of UnetTcn reshapes [N, CH, F, ld] to [N, CH*F, ld] -- the same
memory -- and runs the Conv-TasNet kernels.  stride_t must be 1 (every recipe); norm_type bN2d.
"""
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from .. import hip
from ..ops import op_module, same_shape
from ._plans import PlanCache, _f32
from .conv_tasnet import TCN, GatedTCN, TcnBlock
from .lobe.activation import activation_kind, get_activation
from .._abi import PS_NORM_GLOBAL
from .lobe.norm import GlobLN, get_norm


import os

# 1 (default): Conv2d / ConvTranspose2d as ps_conv2d_f32 (implicit GEMM, taps gathered in the kernel);
# 0: ps_unfold2d_f32 + ps_conv1x1_f32 + ps_activation_f32 (the path gLN-normalised layers always take)
IMPLICIT_CONV = os.environ.get("PS_IMPLICIT_CONV", "1") == "1"
F16X2_STRIDED = os.environ.get("PS_CONV2D_F16X2_STRIDED", "1") == "1"   # 0: stride > 1 transposed layers stay on the fp32 kernel (compacted taps)


def _unet_shape(ctor, x, aux, params):
    """[N, C, F, T] -> the same, or [N, multi_output, C', T] when the up stack emits several outputs (unet.py:253-254)."""
    mo = int((ctor or {}).get("multi_output", 1) or 1)
    if mo == 1:
        return x
    n, t = x[0], x[-1]
    per = 1
    for d in x[1:-1]:
        per *= d
    return (n, mo, per, t)


@op_module("unet_fwd", _unet_shape)
class Unet(PlanCache, nn.Module):
    """unet.py:13-296; constructor order as the reference (unet.py:35-53)."""

    def __init__(self, input_type: str = "RI", input_dim: int = 512, activation_type: str = "PReLU",
                 norm_type: str = "bN2d", dropout: float = 0.05, channels: Tuple = (1, 1, 8, 8, 16, 16),
                 transpose_t_size: int = 2, skip_conv: bool = False, kernel_t: Tuple = (5, 1, 9, 1, 1),
                 stride_t: Tuple = (1, 1, 1, 1, 1), dilation_t: Tuple = (1, 1, 1, 1, 1),
                 kernel_f: Tuple = (1, 5, 1, 5, 1), stride_f: Tuple = (1, 4, 1, 4, 1),
                 dilation_f: Tuple = (1, 1, 1, 1, 1), delay: Tuple = (0, 0, 1, 0, 0), multi_output: int = 1):
        super().__init__()
        assert len(kernel_t) == len(kernel_f) == len(stride_t) == len(stride_f) == len(dilation_t) == len(dilation_f)
        self.input_type = input_type
        self.input_dim = input_dim
        self.multi_output = multi_output
        self.activation_type = activation_type
        self.norm_type = norm_type
        self.dropout = dropout
        self.skip_conv = skip_conv
        self.kernel_t, self.kernel_f = kernel_t, kernel_f
        self.stride_t, self.stride_f = stride_t, stride_f
        self.dilation_t, self.dilation_f = dilation_t, dilation_f
        self.transpose_t_size = transpose_t_size
        active_cls = get_activation(activation_type.lower())
        norm_cls = get_norm(norm_type)
        self.n_cnn = len(kernel_t)
        self.channels = list(channels)
        self.kernel = list(zip(kernel_f, kernel_t))
        self.delay = delay
        self.dilation = list(zip(dilation_f, dilation_t))
        self.stride = list(zip(stride_f, stride_t))
        self.t_kernel = transpose_t_size
        if input_type.lower() == "ri":
            self.num_freq = input_dim // 2
            self.channels[0] = self.channels[0] * 2
        elif input_type.lower() == "real":
            self.num_freq = input_dim
        else:
            raise TypeError("Input feature type should be RI-concate, RI-stack or Real")

        self.cnn_down = nn.ModuleList()
        for i in range(self.n_cnn):
            freq_pad = (self.kernel[i][0] // 2, self.kernel[i][0] // 2)
            time_pad = (self.kernel[i][1] - self.delay[i] - 1, self.delay[i])
            self.cnn_down.append(nn.Sequential(
                nn.ZeroPad2d(time_pad + freq_pad),
                nn.Conv2d(self.channels[i], self.channels[i + 1], kernel_size=self.kernel[i], stride=self.stride[i],
                          dilation=self.dilation[i]),
                norm_cls(self.channels[i + 1]), active_cls(), nn.Dropout(self.dropout)))

        self.cnn_up = nn.ModuleList()
        skip_double = 2 if not skip_conv else 1
        for i in reversed(range(self.n_cnn)):
            s, _ = self.stride[i]
            k = self.kernel[i][0]
            p = k // 2
            op = s - k + 2 * p
            if i != 0:
                layers = [nn.ConvTranspose2d(self.channels[i + 1] * skip_double, self.channels[i],
                                             kernel_size=(k, self.t_kernel), stride=self.stride[i],
                                             dilation=self.dilation[i], padding=(p, 0), output_padding=(op, 0)),
                          norm_cls(self.channels[i]), active_cls()]
            else:
                layers = [nn.ConvTranspose2d(self.channels[i + 1] * skip_double, self.channels[i] * self.multi_output,
                                             kernel_size=(k, self.t_kernel), stride=self.stride[i],
                                             dilation=self.dilation[i], padding=(p, 0), output_padding=(op, 0))]
            self.cnn_up.append(nn.Sequential(*layers))
        if skip_conv:
            self.skip_cnn = nn.ModuleList()
            for i in reversed(range(self.n_cnn)):
                self.skip_cnn.append(nn.Sequential(
                    nn.Conv2d(self.channels[i + 1], self.channels[i + 1], kernel_size=(1, 1), stride=1), active_cls()))

    # -- plan ---------------------------------------------------------------------------------------------------
    @staticmethod
    def _fold(norm: Optional[nn.Module], w2: torch.Tensor, b: torch.Tensor):
        """eval BatchNorm2d after a conv is an affine map of its output: fold it into the GEMM weight / bias."""
        if norm is None or isinstance(norm, GlobLN):
            return w2, b     # gLN needs the statistics of the conv output: applied by ps_norm_activation_f32
        if not isinstance(norm, nn.BatchNorm2d):
            raise NotImplementedError(f"U-Net convolution norm {type(norm).__name__}: only bN2d is on the HIP path")
        if norm.training:
            raise RuntimeError("BatchNorm2d must be in eval() mode on the HIP inference path")
        scale = norm.weight.detach().float() / torch.sqrt(norm.running_var.detach().float() + norm.eps)
        shift = norm.bias.detach().float() - norm.running_mean.detach().float() * scale
        return w2 * scale.to(w2.device).reshape(-1, 1), b * scale.to(b.device) + shift.to(b.device)

    @staticmethod
    def _gln(norm: Optional[nn.Module], bias: torch.Tensor, device) -> dict:
        """gLN parameters + the bias moments that take the pad columns out of the GEMM's statistics (one host read
        at plan time)."""
        if not isinstance(norm, GlobLN):
            return {}
        return dict(gln=(_f32(norm.gamma, device), _f32(norm.beta, device), float(norm.eps)),
                    bias_sum=float(bias.double().sum().item()), bias_sq=float((bias.double() ** 2).sum().item()))

    def _act(self, mod: Optional[nn.Module], device):
        if mod is None:
            return "none", None
        kind = activation_kind(mod)
        if kind == "prelu" and mod.weight.numel() != 1:
            raise NotImplementedError("PReLU with per-channel slopes is not on the HIP path")
        return kind, (_f32(mod.weight, device) if kind == "prelu" else None)

    def _build_unet(self, device):
        if any(st != 1 for _, st in self.stride):
            raise NotImplementedError("U-Net on HIP: stride_t must be 1 (it is in every recipe)")
        if self.training and self.dropout > 0:
            raise RuntimeError("Unet: dropout is active; the HIP path is inference only -- call .eval()")
        down, up, skipc = [], [], []
        for i, seq in enumerate(self.cnn_down):
            conv = seq[1]
            w2, b = self._fold(seq[2], _f32(conv.weight, device).reshape(conv.out_channels, -1),
                               _f32(conv.bias, device))
            kind, slope = self._act(seq[3], device)
            down.append(dict(wt=hip.pack_wt(w2), w2=w2.contiguous(), bias=b.contiguous(), M=conv.out_channels, act=kind,
                             slope=slope, **self._gln(seq[2], b, device)))
        for j, seq in enumerate(self.cnn_up):
            conv = seq[0]
            w = _f32(conv.weight, device)                                      # [Cin, Cout, kf, kt]
            w2 = w.permute(1, 0, 2, 3).reshape(w.shape[1], -1)
            w2, b = self._fold(seq[1] if len(seq) > 1 else None, w2, _f32(conv.bias, device))
            kind, slope = self._act(seq[2] if len(seq) > 2 else None, device)
            up.append(dict(wt=hip.pack_wt(w2.contiguous()), w2=w2.contiguous(), bias=b.contiguous(), M=w.shape[1], act=kind,
                           slope=slope, **self._gln(seq[1] if len(seq) > 1 else None, b, device)))
        if self.skip_conv:
            for seq in self.skip_cnn:
                kind, slope = self._act(seq[1], device)
                skipc.append(dict(wt=hip.pack_wt(_f32(seq[0].weight, device)[:, :, 0, 0]), bias=_f32(seq[0].bias, device),
                                  M=seq[0].out_channels, act=kind, slope=slope))
        return dict(down=down, up=up, skip=skipc)

    # -- pieces -------------------------------------------------------------------------------------------------
    def _gemm_act(self, taps: torch.Tensor, lay: dict, f_out: int, ld: int, t: int, res=None) -> torch.Tensor:
        n = taps.shape[0]
        gln = lay.get("gln")
        y, stats = hip.conv1x1(taps, f_out * ld, lay["wt"], lay["M"], None, lay["bias"], res=res,
                               want_stats=gln is not None,
                               out=torch.empty(n, lay["M"], f_out * ld, dtype=torch.float32, device=taps.device))
        y = y.view(n, lay["M"], f_out, ld)
        if gln is None:
            return hip.activation_(y, lay["act"], lay["slope"], t)
        pro = hip.make_prologue(PS_NORM_GLOBAL, False, stats, float(lay["M"] * f_out * t), gln[2], gln[0], gln[1], None)
        pad = float(f_out * (ld - t))
        return hip.norm_activation_(y, t, pro, lay["bias_sum"] * pad, lay["bias_sq"] * pad, lay["act"], lay["slope"])

    def _conv(self, x, x2, lay: dict, t: int, f_out: int, kf: int, kt: int, sf: int, df: int, dt: int, pf: int, pt: int,
              transposed: bool, t_in=None):
        """One implicit-GEMM convolution of the stack with its epilogue (activation, or the statistics of a gLN that follows):
        the fp16x2 kernel when the module's arithmetic is "fp16x2" and the layer suits it (K >= 32, more than 4 output
        channels, no stride-2 transposed taps to compact), else the exact-fp32 kernels."""
        gln = "gln" in lay
        k = (x.shape[1] + (0 if x2 is None else x2.shape[1])) * kf * kt
        if self.gemm_precision == "fp16x2" and k >= 32 and lay["M"] > 4 and (F16X2_STRIDED or not (transposed and sf > 1)):
            if "f16x2" not in lay:
                lay["f16x2"] = hip.pack_conv2d_f16x2(lay["w2"])
            img, w_exp = lay["f16x2"]
            if gln:
                y, stats = hip.conv2d_f16x2(x, x2, img, w_exp, lay["bias"], lay["M"], t, f_out, kf, kt, sf, df, dt, pf, pt,
                                            transposed, t_in=t_in, want_stats=True)
                return self._gln_act(y, stats, lay, f_out, t)
            return hip.conv2d_f16x2(x, x2, img, w_exp, lay["bias"], lay["M"], t, f_out, kf, kt, sf, df, dt, pf, pt, transposed,
                                    lay["act"], lay["slope"], t_in=t_in)
        if gln:
            y, stats = hip.conv2d_stats(x, x2, lay["wt"], lay["bias"], lay["M"], t, f_out, kf, kt, sf, df, dt, pf, pt, transposed,
                                        t_in=t_in)
            return self._gln_act(y, stats, lay, f_out, t)
        return hip.conv2d(x, x2, lay["wt"], lay["bias"], lay["M"], t, f_out, kf, kt, sf, df, dt, pf, pt, transposed, lay["act"],
                          lay["slope"], t_in=t_in)

    def _gln_act(self, y: torch.Tensor, stats: torch.Tensor, lay: dict, f_out: int, t: int) -> torch.Tensor:
        """gLN + activation behind a convolution whose statistics cover the t valid frames only (no pad-column correction)."""
        gln = lay["gln"]
        pro = hip.make_prologue(PS_NORM_GLOBAL, False, stats, float(lay["M"] * f_out * t), gln[2], gln[0], gln[1], None)
        return hip.norm_activation_(y, t, pro, 0.0, 0.0, lay["act"], lay["slope"])

    def _down(self, x4: torch.Tensor, t: int, p: dict) -> List[torch.Tensor]:
        """[N, CH0, F, ld] -> skip list (input first), unet.py:235-246."""
        skip = [x4]
        x = x4
        for i, lay in enumerate(p["down"]):
            kf, kt = self.kernel[i]
            sf, _ = self.stride[i]
            df, dt = self.dilation[i]
            f_in = x.shape[2]
            pf = kf // 2
            f_out = (f_in + 2 * pf - df * (kf - 1) - 1) // sf + 1
            if IMPLICIT_CONV and x.shape[1] * kf * kt <= 4096:
                # (a gLN behind the convolution gets its statistics from the convolution's epilogue: round 3 built the tap
                #  matrix for these layers, 43 + 47 ms per forward of tse_unet_tcn_v0 at 32 x 4 s)
                x = self._conv(x, None, lay, t, f_out, kf, kt, sf, df, dt, pf, kt - self.delay[i] - 1, False)
            else:
                taps = hip.unfold2d(x, None, t, f_out, kf, kt, sf, df, dt, pf, kt - self.delay[i] - 1, False)
                x = self._gemm_act(taps, lay, f_out, x.shape[3], t)
            skip.append(x)
        return skip

    def _up(self, x: torch.Tensor, skip: List[torch.Tensor], t: int, p: dict, transpose_delay: bool) -> torch.Tensor:
        """CNN-up stack (unet.py:248-256, 514-528) -> [N, CH0*multi_output, F, ld]."""
        for j, i in enumerate(reversed(range(self.n_cnn))):
            lay = p["up"][j]
            s = skip[-j - 1]
            n, _, f_in, ld = x.shape
            if self.skip_conv:
                sl = p["skip"][j]
                sc, _ = hip.conv1x1(s.view(n, s.shape[1], f_in * ld), f_in * ld, sl["wt"], sl["M"], None, sl["bias"],
                                    out=torch.empty(n, sl["M"], f_in * ld, dtype=torch.float32, device=x.device))
                sc = hip.activation_(sc.view(n, sl["M"], f_in, ld), sl["act"], sl["slope"], t)
                x = hip.add_(sc, x)
                x2 = None
            else:
                x2 = s
            sf, _ = self.stride[i]
            kf = self.kernel[i][0]
            df, dt = self.dilation[i]
            pf = kf // 2
            op = sf - kf + 2 * pf
            f_out = (f_in - 1) * sf - 2 * pf + df * (kf - 1) + op + 1
            ext = (self.t_kernel - 1) * dt                       # frames the transposed convolution adds
            if "gln" in lay and ext > 0:
                # the reference normalises the UNtrimmed output (unet.py:252-256 trims after the layer): produce all
                # T + ext frames, normalise over them, then drop the trimmed ones
                if t + ext > ld:
                    raise NotImplementedError("U-Net on HIP: gLN decoder needs rows with room for the untrimmed frames")
                if IMPLICIT_CONV and (x.shape[1] + (0 if x2 is None else x2.shape[1])) * kf * self.t_kernel <= 4096:
                    x = self._conv(x, x2, lay, t + ext, f_out, kf, self.t_kernel, sf, df, dt, pf, 0, True, t_in=t)
                else:
                    taps = hip.unfold2d(x, x2, t + ext, f_out, kf, self.t_kernel, sf, df, dt, pf, 0, True, t_in=t)
                    x = self._gemm_act(taps, lay, f_out, ld, t + ext)
                if transpose_delay:   # keep frames [ext, ext + T): a 1x1 "convolution" with a negative time pad
                    x = hip.unfold2d(x, None, t, f_out, 1, 1, 1, 1, 1, 0, -ext, False, t_in=t + ext).view(
                        n, lay["M"], f_out, ld)
            elif IMPLICIT_CONV and (x.shape[1] + (0 if x2 is None else x2.shape[1])) * kf * self.t_kernel <= 4096:
                x = self._conv(x, x2, lay, t, f_out, kf, self.t_kernel, sf, df, dt, pf, ext if transpose_delay else 0, True)
            else:
                shift = ext if transpose_delay else 0
                taps = hip.unfold2d(x, x2, t, f_out, kf, self.t_kernel, sf, df, dt, pf, shift, True)
                x = self._gemm_act(taps, lay, f_out, ld, t)
        return x

    def frames_needed(self, t: int) -> int:
        """Row length the decoder needs: gLN layers hold their untrimmed T + transpose_t_size - 1 frames."""
        return t + (self.t_kernel - 1) * max(dt for _, dt in self.dilation)

    def _split_in(self, x: torch.Tensor):
        """[N, C, T] compact -> ([N, CH0, F, ld], T)."""
        t = x.shape[-1]
        xp = hip.pad_rows(x if x.dim() == 3 else x.reshape(x.shape[0], -1, t), self.frames_needed(t))
        n, c, ld = xp.shape
        ch0 = 2 if self.input_type.lower() == "ri" else 1
        return xp.view(n, ch0, c // ch0, ld), t

    def _merge_out(self, y: torch.Tensor, t: int) -> torch.Tensor:
        """[N, CH0*mo, F, ld] -> the reference's output shape (unet.py:258-283)."""
        n, c, f, ld = y.shape
        out = hip.unpad_rows(y.reshape(n, c * f, ld), t)
        if self.multi_output != 1:
            return out.reshape(n, self.multi_output, -1, t)
        return out

    def _rows4(self, x_pad: torch.Tensor) -> torch.Tensor:
        n, c, ld = x_pad.shape
        ch0 = 2 if self.input_type.lower() == "ri" else 1
        return x_pad.view(n, ch0, c // ch0, ld)

    def forward_padded4(self, x4: torch.Tensor, t: int, dvec: Optional[torch.Tensor] = None) -> torch.Tensor:
        p = self._plan_get(x4.device, self._build_unet)
        skip = self._down(x4, t, p)
        return self._up(skip[-1], skip, t, p, False)

    def forward_padded(self, x_pad: torch.Tensor, t: int, dvec: Optional[torch.Tensor] = None,
                       lane: int = 0) -> torch.Tensor:
        """Wrapper entry: padded [N, C, ld] -> padded [N, C, ld] (RI halves are the two CH0 planes: no data moves)."""
        if self.multi_output != 1:
            raise NotImplementedError("multi_output U-Net behind the single-output wrapper")
        y = self.forward_padded4(self._rows4(x_pad), t, dvec)
        return y.reshape(y.shape[0], -1, y.shape[3])

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """x [N, C, T] -> [N, C, T] (unet.py:221-283)."""
        hip.require_device(x, "Unet.forward")
        x4, t = self._split_in(x)
        return self._merge_out(self.forward_padded4(x4, t), t)

    @property
    def get_args(self) -> Dict:
        return {"input_type": self.input_type, "input_dim": self.input_dim, "activation_type": self.activation_type,
                "norm_type": self.norm_type, "dropout": self.dropout, "channels": self.channels,
                "transpose_t_size": self.transpose_t_size, "skip_conv": self.skip_conv, "kernel_t": self.kernel_t,
                "stride_t": self.stride_t, "dilation_t": self.dilation_t, "kernel_f": self.kernel_f,
                "stride_f": self.stride_f, "dilation_f": self.dilation_f, "delay": self.delay,
                "multi_output": self.multi_output}


@op_module("unet_tcn_fwd", _unet_shape)
class UnetTcn(Unet):
    """U-Net with a (gated) TCN bottleneck (unet.py:298-557)."""

    def __init__(self, embed_dim: int = 0, embed_norm: bool = False, input_type: str = "RI", input_dim: int = 512,
                 activation_type: str = "PReLU", norm_type: str = "bN2d", dropout: float = 0.05,
                 channels: Tuple = (1, 1, 8, 8, 16, 16), transpose_t_size: int = 2, transpose_delay: bool = False,
                 skip_conv: bool = False, kernel_t: Tuple = (5, 1, 9, 1, 1), stride_t: Tuple = (1, 1, 1, 1, 1),
                 dilation_t: Tuple = (1, 1, 1, 1, 1), kernel_f: Tuple = (1, 5, 1, 5, 1),
                 stride_f: Tuple = (1, 4, 1, 4, 1), dilation_f: Tuple = (1, 1, 1, 1, 1), delay: Tuple = (0, 0, 1, 0, 0),
                 tcn_layer: str = "normal", tcn_kernel: int = 3, tcn_dim: int = 256, tcn_dilated_basic: int = 2,
                 per_tcn_stack: int = 5, repeat_tcn: int = 4, tcn_with_embed: List = [1, 0, 0, 0, 0],
                 tcn_use_film: bool = False, tcn_norm: str = "gLN", dconv_norm: str = "gGN", causal: bool = False):
        super().__init__(input_type, input_dim, activation_type, norm_type, dropout, channels, transpose_t_size,
                         skip_conv, kernel_t, stride_t, dilation_t, kernel_f, stride_f, dilation_f, delay)
        self.embed_dim = embed_dim
        self.embed_norm = embed_norm
        self.tcn_layer = tcn_layer
        self.tcn_dim = tcn_dim
        self.tcn_kernel = tcn_kernel
        self.per_tcn_stack = per_tcn_stack
        self.repeat_tcn = repeat_tcn
        self.tcn_dilated_basic = tcn_dilated_basic
        self.tcn_with_embed = tcn_with_embed
        self.tcn_norm = tcn_norm
        self.dconv_norm = dconv_norm
        self.tcn_use_film = tcn_use_film
        self.causal = causal
        self.transpose_delay = transpose_delay

        dim = self.num_freq
        for stride, _ in self.stride:
            dim = dim // stride if dim % stride == 0 else dim // stride + 1
        dim *= self.channels[-1]
        self.temporal_input_dim = dim
        if self.tcn_layer.lower() == "normal":
            tcn_cls = TCN
        elif self.tcn_layer.lower() == "gated":
            print("GatedTCN would ignore dconv_norm configuration.")
            tcn_cls = GatedTCN
        else:
            raise NameError
        assert per_tcn_stack == len(tcn_with_embed)
        self.tcn_list = nn.ModuleList()
        for _ in range(repeat_tcn):
            stack = []
            for i in range(per_tcn_stack):
                kw = dict(kernel=tcn_kernel, dilation=tcn_dilated_basic ** i,
                          emb_dim=embed_dim if tcn_with_embed[i] else 0, causal=causal, tcn_norm=tcn_norm)
                if tcn_cls is TCN:
                    kw["dconv_norm"] = dconv_norm
                else:
                    kw["use_film"] = tcn_use_film if tcn_with_embed[i] else False
                stack.append(tcn_cls(dim, tcn_dim, **kw))
            self.tcn_list.append(nn.ModuleList(stack))

    def _bottleneck(self, y: torch.Tensor, t: int, dvec: Optional[torch.Tensor]) -> torch.Tensor:
        """[N, CH*F, ld] through the TCN stack (unet.py:489-497)."""
        if dvec is not None and self.embed_norm:
            dvec = hip.l2_normalize(dvec.float())
        blocks = [m for stack in self.tcn_list for m in stack]
        embeds = [dvec if (self.tcn_with_embed[i % self.per_tcn_stack] and dvec is not None) else None
                  for i in range(len(blocks))]
        if self.tcn_layer.lower() == "normal" and all(m.plan(y.device)["fused"] for m in blocks):
            arr = (TcnBlock * len(blocks))(*[m.plan(y.device)["block"] for m in blocks])
            if any(self.tcn_with_embed) and dvec is None:
                raise RuntimeError("UnetTcn.forward: tcn_with_embed is set but no dvec was given")
            return hip.conv_tasnet(arr, len(blocks), y, t, self.temporal_input_dim, self.tcn_dim, dvec, False)
        for m, e in zip(blocks, embeds):
            y = m.forward_padded(y, t, e) if isinstance(m, GatedTCN) else m.forward_padded_staged(y, t, e)
        return y

    def forward_padded4(self, x4: torch.Tensor, t: int, dvec: Optional[torch.Tensor] = None) -> torch.Tensor:
        p = self._plan_get(x4.device, self._build_unet)
        skip = self._down(x4, t, p)
        n, ch, f, ld = skip[-1].shape
        y = self._bottleneck(skip[-1].reshape(n, ch * f, ld), t, dvec)
        return self._up(y.view(n, ch, f, ld), skip, t, p, self.transpose_delay)

    def forward(self, x: torch.Tensor, dvec: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x [N, C, T], dvec [N, E] -> [N, C, T] (unet.py:454-537)."""
        hip.require_device(x, "UnetTcn.forward")
        x4, t = self._split_in(x)
        return self._merge_out(self.forward_padded4(x4, t, dvec), t)

    @property
    def get_args(self) -> Dict:
        a = dict(Unet.get_args.fget(self))
        a.pop("multi_output")
        a.update(transpose_delay=self.transpose_delay, embed_dim=self.embed_dim, embed_norm=self.embed_norm,
                 tcn_norm=self.tcn_norm, dconv_norm=self.dconv_norm, tcn_layer=self.tcn_layer, tcn_dim=self.tcn_dim,
                 tcn_kernel=self.tcn_kernel, tcn_dilated_basic=self.tcn_dilated_basic, repeat_tcn=self.repeat_tcn,
                 per_tcn_stack=self.per_tcn_stack, tcn_with_embed=self.tcn_with_embed, tcn_use_film=self.tcn_use_film,
                 causal=self.causal)
        return a
