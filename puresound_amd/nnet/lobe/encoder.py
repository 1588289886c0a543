"""Encoder / decoder filterbanks (mirror of puresound/nnet/lobe/encoder.py:16-183, 275-456).

FreeEncDec runs on ps_free_encode_f32 / ps_free_decode_f32.  ConvEncDec / ConvSTFT carry the
reference's parameters and buffers (wsin, wcos, kernel_{sin,cos}_inv, window_mask) under the same
state_dict keys; their kernels are the next row of the hot-path table.
"""
import math
from typing import Optional

import torch
import torch.nn as nn

from ... import hip


class FreeEncDec(nn.Module):
    """Free (learned) filters: waveform -> latent feats -> waveform (encoder.py:16-94)."""

    def __init__(self, win_length: int = 512, laten_length: int = 512, hop_length: int = 128,
                 output_active: bool = False):
        super().__init__()
        self.win_length = win_length
        self.hop_length = hop_length
        self.output_active = output_active
        # parameter holders with the reference's keys encoder.weight / decoder.weight ([C,1,win])
        self.encoder = nn.Conv1d(1, laten_length, kernel_size=win_length, stride=hop_length, bias=False)
        self.decoder = nn.ConvTranspose1d(laten_length, 1, kernel_size=win_length, stride=hop_length, bias=False)

    # -- padded-layout entry points used by the fused wrapper ---------------------------------
    def encode_padded(self, x: torch.Tensor):
        """[N,L] -> (padded feats [N,C,ldt], T)."""
        return hip.free_encode(x, self.encoder.weight.detach(), self.hop_length, self.output_active)

    def decode_padded(self, feats_pad: torch.Tensor, t: int, mask_pad: Optional[torch.Tensor] = None,
                      mask_act: str = "linear", out_mode: str = "none",
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
        return hip.free_decode(feats_pad, t, self.decoder.weight.detach(), self.hop_length, mask_pad, mask_act,
                               out_mode, out)

    # -- reference API -------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[N,L] -> [N,C,T] (encoder.py:71-83)."""
        feats, t = self.encode_padded(x)
        return hip.unpad_rows(feats, t)

    def inverse(self, x: torch.Tensor) -> torch.Tensor:
        """[N,C,T] -> [N,L] (encoder.py:85-94)."""
        hip.require_device(x, "FreeEncDec.inverse")
        return self.decode_padded(hip.pad_rows(x), x.shape[-1])


def create_fourier_kernels(n_fft: int):
    """float64 sin/cos tables -> fp32, freq_scale="no" (lobe/stft.py:91-100): [n_fft/2+1, 1, n_fft]."""
    s = torch.arange(n_fft, dtype=torch.float64)
    k = torch.arange(n_fft // 2 + 1, dtype=torch.float64).reshape(-1, 1)
    ang = 2 * math.pi * k * s / n_fft
    return torch.sin(ang).to(torch.float32).unsqueeze(1), torch.cos(ang).to(torch.float32).unsqueeze(1)


class ConvSTFT(nn.Module):
    """Conv-STFT with trainable analysis kernels (encoder.py:275-456).  Parameter / buffer holder."""

    def __init__(self, window_mask: torch.Tensor, n_fft: int = 2048, win_length: Optional[int] = None,
                 freq_bins: Optional[int] = None, hop_length: Optional[int] = None, freq_scale: str = "no",
                 iSTFT: bool = False, fmin: int = 50, fmax: int = 6000, sr: int = 22050, trainable: bool = False,
                 output_format: str = "Complex"):
        super().__init__()
        if win_length is None:
            win_length = n_fft
        if hop_length is None:
            hop_length = int(win_length // 4)
        if freq_scale != "no":
            raise NotImplementedError("only freq_scale='no' (every recipe's setting) is mirrored")
        self.output_format = output_format
        self.trainable = trainable
        self.stride = hop_length
        self.n_fft = n_fft
        self.freq_bins = freq_bins
        self.win_length = win_length
        self.iSTFT = iSTFT
        kernel_sin, kernel_cos = create_fourier_kernels(n_fft)
        if iSTFT:
            # conjugate-extended inverse kernels, not windowed, never trainable (encoder.py:329-335)
            self.register_buffer("kernel_sin_inv",
                                 torch.cat((kernel_sin, -kernel_sin[1:-1].flip(0)), 0).unsqueeze(-1))
            self.register_buffer("kernel_cos_inv",
                                 torch.cat((kernel_cos, kernel_cos[1:-1].flip(0)), 0).unsqueeze(-1))
        if len(window_mask) != self.n_fft:
            raise TypeError("only support window length == n_fft")
        wsin = kernel_sin * window_mask
        wcos = kernel_cos * window_mask
        if self.trainable:
            self.register_parameter("wsin", nn.Parameter(wsin, requires_grad=True))
            self.register_parameter("wcos", nn.Parameter(wcos, requires_grad=True))
        else:
            self.register_buffer("wsin", wsin)
            self.register_buffer("wcos", wcos)
        self.register_buffer("window_mask", window_mask.unsqueeze(0).unsqueeze(-1))

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        raise NotImplementedError("ConvSTFT.forward: the conv-STFT HIP kernel is the next hot-path row")

    def inverse(self, X: torch.Tensor, refresh_win: bool = True) -> torch.Tensor:
        if not hasattr(self, "kernel_sin_inv") or not hasattr(self, "kernel_cos_inv"):
            raise NameError("Please activate the iSTFT module by setting `iSTFT=True` if you want to use `inverse`")
        assert X.dim() == 4, "Inverse iSTFT only works for complex number (batch, freq_bins, timesteps, 2)."
        raise NotImplementedError("ConvSTFT.inverse: the iSTFT HIP kernel is the next hot-path row")


class ConvEncDec(nn.Module):
    """STFT encoder/decoder facade (encoder.py:97-183)."""

    def __init__(self, fft_length: int = 512, win_type: str = "hann", win_length: int = 512,
                 freq_bins: int = None, hop_length: int = 128, freq_scale: str = "no", iSTFT: bool = True,
                 fmin: int = 0, fmax: int = 8000, sr: int = 16000, trainable: bool = True,
                 output_format: str = "Complex"):
        super().__init__()
        self.n_fft = fft_length
        self.win_length = win_length
        self.freq_bins = freq_bins
        self.hop_length = hop_length
        self.freq_scale = freq_scale
        self.iSTFT = iSTFT
        self.fmin, self.fmax, self.sr = fmin, fmax, sr
        self.trainable = trainable
        self.output_format = output_format
        self.window = self.get_windows(win_type)
        self.encoder = ConvSTFT(self.window, n_fft=self.n_fft, win_length=self.win_length,
                                freq_scale=self.freq_scale, iSTFT=self.iSTFT, sr=self.sr, fmin=self.fmin,
                                fmax=self.fmax, output_format=self.output_format, trainable=self.trainable,
                                hop_length=self.hop_length)

    def get_windows(self, type: str) -> torch.Tensor:
        if type.lower() == "hann":
            return torch.hann_window(self.win_length)
        raise NotImplementedError("window type not support")

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.encoder(x.unsqueeze(1))

    def inverse(self, x: torch.Tensor) -> torch.Tensor:
        gen = self.encoder.inverse(x)
        return gen.squeeze(1) if gen.dim() == 3 else gen
