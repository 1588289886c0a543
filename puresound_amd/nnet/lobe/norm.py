"""Normalisation lobes (mirror of puresound/nnet/lobe/norm.py:5-112).

On the HIP path a norm is never a kernel of its own: its statistics are produced by the epilogue of
the convolution in front of it and it is applied in the prologue of the convolution behind it
(csrc/conv1x1.hip, csrc/dwconv.hip).  These classes therefore only hold the parameters, under the
reference's state_dict keys, and say how the kernels should treat them.
"""
import torch
import torch.nn as nn

from ..._abi import PS_NORM_AFFINE, PS_NORM_GLOBAL


class _LayerNorm(nn.Module):
    """Parameter holder with the reference's keys `gamma` / `beta` (norm.py:5-17)."""

    def __init__(self, channel_size):
        super().__init__()
        self.eps = 1e-8
        self.channel_size = channel_size
        self.gamma = nn.Parameter(torch.ones(channel_size), requires_grad=True)
        self.beta = nn.Parameter(torch.zeros(channel_size), requires_grad=True)

    def forward(self, x):
        raise NotImplementedError(
            f"{type(self).__name__} is fused into the neighbouring convolution kernels on the HIP path; "
            "it is not callable on its own")


class GlobLN(_LayerNorm):
    """gLN: per-utterance statistics over [C,T] (norm.py:20-34)."""


class ChanLN(_LayerNorm):
    """cLN: per-frame statistics over C (norm.py:37-50).  Parameters only; no HIP kernel yet."""


class InstantLN(_LayerNorm):
    """iLN (norm.py:53-68).  Parameters only; off the Conv-TasNet path."""


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
