"""Normalisation lobes (mirror of puresound/nnet/lobe/norm.py:5-112).

Inside the models a norm is never a kernel of its own: its statistics are produced by the epilogue of
the convolution in front of it and it is applied in the prologue of the convolution behind it
(csrc/conv1x1.hip, csrc/dwconv.hip).  Called on their own the classes run: GlobLN = ps_row_stats_f64 +
ps_norm_activation_f32, ChanLN / InstantLN = ps_chan_layernorm_f32.
"""
import torch
import torch.nn as nn

from ... import hip
from ..._abi import PS_NORM_AFFINE, PS_NORM_GLOBAL


class _LayerNorm(nn.Module):
    """Parameter holder with the reference's keys `gamma` / `beta` (norm.py:5-17)."""

    def __init__(self, channel_size):
        super().__init__()
        self.eps = 1e-8
        self.channel_size = channel_size
        self.gamma = nn.Parameter(torch.ones(channel_size), requires_grad=True)
        self.beta = nn.Parameter(torch.zeros(channel_size), requires_grad=True)

    def _params(self, x):
        f32 = dict(dtype=torch.float32, device=x.device)
        return self.gamma.detach().to(**f32).contiguous(), self.beta.detach().to(**f32).contiguous()

    def _per_position(self, x: torch.Tensor, channels: int) -> torch.Tensor:
        """statistics over `channels` (dim 1 of the [N, channels, positions] view) at every position"""
        hip.require_device(x, type(self).__name__ + ".forward")
        with torch.no_grad():
            g, b = self._params(x)
            if g.numel() != channels:
                raise RuntimeError(f"{type(self).__name__}: {channels} channels, {g.numel()} gains")
            rows = x.float().reshape(x.shape[0], channels, -1)
            t = rows.shape[-1]
            return hip.unpad_rows(hip.chan_layernorm(hip.pad_rows(rows), t, g, b, self.eps), t).reshape(x.shape)


class GlobLN(_LayerNorm):
    """gLN: per-utterance statistics over everything but the batch axis (norm.py:20-34)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[N, C, *] -> same shape: (x - mean) / sqrt(var + eps) * gamma[c] + beta[c]."""
        hip.require_device(x, "GlobLN.forward")
        if x.dim() < 3:
            raise RuntimeError("GlobLN: input of at least three dimensions [batch, chan, *]")
        with torch.no_grad():
            g, b = self._params(x)
            n, c, t = x.shape[0], x.shape[1], x.shape[-1]
            inner = x[0, 0].numel() // t                      # rows per channel
            rows = hip.pad_rows(x.float().reshape(n, c * inner, t))
            stats = hip.row_stats(rows, t)
            pro = hip.make_prologue(PS_NORM_GLOBAL, False, stats, float(c * inner * t), self.eps, g, b, None)
            hip.norm_activation_(rows.view(n, c, inner, rows.shape[-1]), t, pro, 0.0, 0.0, "none", None)
            return hip.unpad_rows(rows, t).reshape(x.shape)


class ChanLN(_LayerNorm):
    """cLN: statistics over the channel axis at every position (norm.py:37-50)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self._per_position(x, x.shape[1])


class InstantLN(_LayerNorm):
    """iLN (norm.py:53-68): [N, CH, C, T], statistics over CH * C at every frame."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 4:
            raise ValueError("InstantLN: [N, CH, C, T] input")  # (the reference fails unpacking the shape)
        n, ch, c, t = x.shape
        return self._per_position(x.reshape(n, ch * c, t), ch * c).reshape(n, ch, c, t)


# Aliases, as norm.py:90-97
gLN = GlobLN
cLN = ChanLN
iLN = InstantLN
bN1d = nn.BatchNorm1d
bN2d = nn.BatchNorm2d
gGN = lambda x: nn.GroupNorm(1, x, 1e-8)  # noqa: E731


def get_norm(name: str):
    """Same contract as norm.py:100-112: NameError for anything outside the six identifiers."""
    if name not in ["gLN", "cLN", "iLN", "bN1d", "gGN", "bN2d"]:
        raise NameError("Could not interpret normalization identifier")
    return globals()[name]


def norm_plan(mod: nn.Module):
    """(PS_NORM_* kind, gain[K], shift[K]) the kernels consume for a norm module, fp32 detached.

    GlobLN / GroupNorm(1): kind GLOBAL, (gamma, beta).  BatchNorm1d in eval mode is folded to a
    per-channel scale/shift: kind AFFINE."""
    if isinstance(mod, GlobLN):
        return PS_NORM_GLOBAL, mod.gamma.detach().float(), mod.beta.detach().float()
    if isinstance(mod, nn.GroupNorm):
        if mod.num_groups != 1:
            raise NotImplementedError("only GroupNorm(1, C) (gGN) is on the HIP path")
        return PS_NORM_GLOBAL, mod.weight.detach().float(), mod.bias.detach().float()
    if isinstance(mod, nn.BatchNorm1d):
        if mod.training:
            raise RuntimeError("BatchNorm1d must be in eval() mode on the HIP inference path")
        scale = mod.weight.detach().float() / torch.sqrt(mod.running_var.detach().float() + mod.eps)
        shift = mod.bias.detach().float() - mod.running_mean.detach().float() * scale
        return PS_NORM_AFFINE, scale, shift
    raise NotImplementedError(f"{type(mod).__name__} has no HIP kernel on the TCN path yet (gLN, gGN, bN1d do)")
