"""Signal-level scores of the reference's loss/sdr.py, forward only (evaluation of the HIP path's outputs; the
inference stack has no autograd): `SDRLoss` with the reference's constructor, `init_mode` aliases and `forward`
signature (puresound/nnet/loss/sdr.py:7-215), `si_snr` (:263-299), `inactive_sdr_loss` (:302-322), `l2_norm` (:248-260).

Every variant is algebra on five moments of the (estimate, reference) pair, which `ps_wave_moments_f64` gathers in one
streaming pass over the two waveforms (fp64 accumulation); with a = estimate, b = reference, L samples:
  zero-mean inner products  <a,b> = S_ab - S_a S_b / L,  <a,a> = S_aa - S_a^2 / L,  <b,b> = S_bb - S_b^2 / L
  alpha = <a,b> / (<b,b> + eps)            (scaled target s_target = alpha b, sdr.py:146-149)
  |s_target|^2 = alpha^2 <b,b>,   |a - s_target|^2 = <a,a> - 2 alpha <a,b> + alpha^2 <b,b>   (:151-157)
"""
from typing import Optional

import torch
import torch.nn as nn

from ... import hip


def _moments(s1: torch.Tensor, s2: torch.Tensor) -> torch.Tensor:
    """[..., L] x 2 -> fp64 moments [..., 5] = (S_a, S_b, S_aa, S_bb, S_ab)."""
    lead, length = s1.shape[:-1], s1.shape[-1]
    return hip.wave_moments(s1.reshape(-1, length), s2.reshape(-1, length)).reshape(*lead, 5)


def _inner_of(m: torch.Tensor, length: int, zero_mean: bool):
    """moments [..., 5] over `length` samples -> fp64 (<a,a>, <b,b>, <a,b>) of shape [..., 1] (optionally of the
    zero-mean signals)."""
    sa, sb, saa, sbb, sab = (m[..., i:i + 1] for i in range(5))
    if zero_mean:
        saa = saa - sa * sa / length
        sbb = sbb - sb * sb / length
        sab = sab - sa * sb / length
    return saa, sbb, sab


def _inner(s1: torch.Tensor, s2: torch.Tensor, zero_mean: bool):
    """[..., L] x 2 -> fp64 (<a,a>, <b,b>, <a,b>) of shape [..., 1] (optionally of the zero-mean signals)."""
    return _inner_of(_moments(s1, s2), s1.shape[-1], zero_mean)


def l2_norm(s1: torch.Tensor, s2: torch.Tensor) -> torch.Tensor:
    """sum(s1 * s2, -1, keepdim=True) (sdr.py:248-260)."""
    return _inner(s1, s2, False)[2].float()


def si_snr(s1: torch.Tensor, s2: torch.Tensor, eps: float = 1e-8, reduction: bool = True) -> torch.Tensor:
    """Single-source SI-SNR in dB (sdr.py:263-299)."""
    return si_snr_from_moments(_moments(s1, s2), s1.shape[-1], eps, reduction)


def si_snr_from_moments(m: torch.Tensor, length: int, eps: float = 1e-8, reduction: bool = True) -> torch.Tensor:
    """si_snr of signals given by their moments [..., 5] over `length` samples (hip.wave_moments, or the decoder's
    epilogue: hip.free_decode_moments)."""
    aa, bb, ab = _inner_of(m, length, True)
    alpha = ab / (bb + eps)
    target = alpha * alpha * bb
    noise = (aa - 2 * alpha * ab + alpha * alpha * bb).clamp_min(0.0)
    snr = (10 * torch.log10(target / (noise + eps) + eps)).float()
    return torch.mean(snr) if reduction else snr


def inactive_sdr_loss(s1: torch.Tensor, s2: torch.Tensor, reduction: bool = True) -> torch.Tensor:
    """10 log10(|s1|^2 + 0.01 |s2|^2 + 1e-8) on the zero-mean signals (sdr.py:302-322)."""
    return _inactive_from_moments(_moments(s1, s2), s1.shape[-1], reduction)


def _inactive_from_moments(m: torch.Tensor, length: int, reduction: bool = True) -> torch.Tensor:
    aa, bb, _ = _inner_of(m, length, True)
    val = (10 * torch.log10(aa + 0.01 * bb + 1e-8)).float()
    return torch.mean(val) if reduction else val


class SDRLoss(nn.Module):
    """sdr.py:7-215: same constructor, aliases and forward semantics; returns the NEGATIVE SDR like the reference."""

    def __init__(self, scaled: bool = True, scale_dependent: bool = False, zero_mean: bool = True,
                 source_aggregated: bool = False, sdr_max: int = None, eps: float = 1e-8, reduction: bool = True,
                 threshold: Optional[float] = None) -> None:
        super().__init__()
        self.scaled = scaled
        self.scale_dependent = scale_dependent
        self.zero_mean = zero_mean
        self.source_aggregated = source_aggregated
        self.sdr_max = sdr_max
        self.eps = eps
        self.reduction = reduction
        self.threshold = threshold

    @classmethod
    def init_mode(cls, loss_func: str = "sisnr", reduction: bool = True, threshold: Optional[float] = None):
        loss_func = loss_func.lower()
        if loss_func not in ("sisnr", "sdsdr", "sdr", "tsdr", "sasdr", "sasisnr", "satsdr"):
            raise NameError  # sdr.py:70-71
        # (the reference tests `loss_func in "sdsdr"`, a substring test: "sdr" counts as scaled, sdr.py:73)
        scaled = loss_func == "sisnr" or loss_func in "sdsdr" or loss_func == "sasisdr"
        print(f"init loss function: {loss_func}")
        return cls(scaled=scaled, scale_dependent=loss_func == "sdsdr", zero_mean=True,
                   source_aggregated=loss_func in ("sasdr", "sasisnr", "satsdr"),
                   sdr_max=30 if loss_func in ("tsdr", "satsdr") else None, eps=1e-8, reduction=reduction,
                   threshold=threshold)

    def check_input_shape(self, s: torch.Tensor) -> None:
        if self.source_aggregated:
            assert s.dim() == 3, "source_aggregated need input dimension is 3"
        else:
            assert s.dim() == 2, "need input shape as (batch, length)"

    @torch.no_grad()
    def forward(self, s1: torch.Tensor, s2: torch.Tensor, inactive_labels: Optional[torch.Tensor] = None) -> torch.Tensor:
        self.check_input_shape(s1)
        self.check_input_shape(s2)
        return self.from_moments(_moments(s1, s2), s1.shape[-1], inactive_labels)

    @torch.no_grad()
    def from_moments(self, m: torch.Tensor, length: int, inactive_labels: Optional[torch.Tensor] = None) -> torch.Tensor:
        """forward() on signals given by their moments: m [N, 5] (or [N, M, 5] for the source-aggregated modes) over
        `length` samples, e.g. from the decoder's epilogue (hip.free_decode_moments).  Rows are independent, so the
        reference's split into active / inactive rows (sdr.py:123-136) is a split of the rows of m."""
        inactive_loss = None
        if inactive_labels is not None and bool((inactive_labels == True).any()):  # noqa: E712  (sdr.py:123-136)
            active_idx = torch.where(inactive_labels == False)[0]  # noqa: E712
            inactive_idx = torch.where(inactive_labels == True)[0]  # noqa: E712
            inactive_loss = _inactive_from_moments(m[inactive_idx], length, reduction=False)
            m = m[active_idx]
        if m.shape[0] > 0:
            aa, bb, ab = _inner_of(m, length, self.zero_mean)
            alpha = ab / (bb + self.eps) if self.scaled else torch.ones_like(ab)
            target = alpha * alpha * bb                                  # |s_target|^2
            if not self.scale_dependent:
                noise = aa - 2 * alpha * ab + alpha * alpha * bb         # |s1 - s_target|^2
            else:
                noise = aa - 2 * ab + bb                                 # |s1 - s2|^2
            noise = noise.clamp_min(0.0)
            if self.sdr_max is not None:
                noise = noise + 10 ** (-self.sdr_max / 10) * target
            if not self.source_aggregated:
                snr = 10 * torch.log10(target / (noise + self.eps) + self.eps)
            else:
                # sdr.py:165-169 sums the norms over dim -1, which is the kept (size 1) time axis of [N, M, 1]: the
                # "aggregated" score of the reference is the per-source score with shape [N, M]; mirrored as is
                snr = 10 * torch.log10(target.sum(dim=-1) / (noise.sum(dim=-1) + self.eps) + self.eps)
            snr = (-snr).float()
        else:
            snr = torch.zeros(0, 1, dtype=torch.float32, device=m.device)
        if self.threshold is not None:
            keep = snr[snr > self.threshold]
            if keep.nelement() > 0:
                snr = keep.view(-1, 1)
        if inactive_loss is not None:
            snr = torch.cat([snr, inactive_loss.view(-1, 1)], dim=0)
        return torch.mean(snr) if self.reduction else snr
