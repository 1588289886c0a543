"""Encoder / decoder filterbanks (mirror of puresound/nnet/lobe/encoder.py:16-183, 275-456).

FreeEncDec runs on ps_free_encode_f32 / ps_free_decode_f32.  ConvEncDec / ConvSTFT carry the reference's
parameters and buffers (wsin, wcos, kernel_{sin,cos}_inv, window_mask) under the same state_dict keys; both
transforms are dense products against those (trainable) tables on ps_conv1x1_f32, framed by ps_frame_f32 and
finished by ps_istft_ola_f32.
"""
import math
from typing import Optional

import torch
import torch.nn as nn

from ... import hip
from ...ops import op_module


class _EncoderConv(nn.Conv1d):
    """`FreeEncDec.encoder`: the reference's nn.Conv1d (same state_dict key `weight`), forward on the HIP operator."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # [N,1,L] -> [N,C,T]
        with torch.no_grad():
            return torch.ops.puresound_amd.free_encode(x[:, 0, :], self.weight, self.stride[0], False)


class _DecoderConvT(nn.ConvTranspose1d):
    """`FreeEncDec.decoder`: the reference's nn.ConvTranspose1d; the recipe's export action traces this submodule on
    its own (egs/tse/main.py:436-439)."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:  # [N,C,T] -> [N,1,L]
        with torch.no_grad():
            return torch.ops.puresound_amd.free_decode(x, self.weight, self.stride[0]).unsqueeze(1)


class FreeEncDec(nn.Module):
    """Free (learned) filters: waveform -> latent feats -> waveform (encoder.py:16-94)."""

    def __init__(self, win_length: int = 512, laten_length: int = 512, hop_length: int = 128,
                 output_active: bool = False):
        super().__init__()
        self.win_length = win_length
        self.hop_length = hop_length
        self.output_active = output_active
        # parameter holders with the reference's keys encoder.weight / decoder.weight ([C,1,win])
        self.encoder = _EncoderConv(1, laten_length, kernel_size=win_length, stride=hop_length, bias=False)
        self.decoder = _DecoderConvT(laten_length, 1, kernel_size=win_length, stride=hop_length, bias=False)

    # -- padded-layout entry points used by the fused wrapper ---------------------------------
    def encode_padded(self, x: torch.Tensor, min_frames=None):
        """[N,L] -> (padded feats [N,C,ldt], T); rows hold at least min_frames(T) frames.  Frames in [T, ldt) are
        UNDEFINED (uninitialised memory no kernel reads as data) unless min_frames(T) > T: then they are zero, as the
        segment padding of the dual-path maskers needs them."""
        return hip.free_encode(x, self.encoder.weight.detach(), self.hop_length, self.output_active, min_frames)

    def feature_bound(self, x: torch.Tensor) -> torch.Tensor:
        """[N,L] -> [N,1]: an upper bound on |forward(x)[n]|: max |x[n]| times the largest row sum of |w| (the ReLU only
        shrinks).  Consumers that scale their input into a narrow exponent range (the fp16x2 GEMMs) take it instead of
        measuring the features.  (The row sum is read back to the host once per weight version.)"""
        w = self.encoder.weight
        key = (w.data_ptr(), w._version)
        if getattr(self, "_l1_key", None) != key:
            self._l1 = float(w.detach().abs().sum(dim=(1, 2)).max())
            self._l1_key = key
        return torch.linalg.vector_norm(x.detach().float(), ord=float("inf"), dim=1, keepdim=True) * self._l1

    def decode_padded(self, feats_pad: torch.Tensor, t: int, mask_pad: Optional[torch.Tensor] = None,
                      mask_act: str = "linear", out_mode: str = "none",
                      out: Optional[torch.Tensor] = None) -> torch.Tensor:
        return hip.free_decode(feats_pad, t, self.decoder.weight.detach(), self.hop_length, mask_pad, mask_act,
                               out_mode, out)

    def decode_scored_padded(self, feats_pad: torch.Tensor, t: int, ref: torch.Tensor,
                             mask_pad: Optional[torch.Tensor] = None, mask_act: str = "linear", out_mode: str = "none",
                             out: Optional[torch.Tensor] = None):
        """decode_padded + the moments [N, 5] of (estimate, ref aligned as base_nn.py:398-412) from the same launch."""
        return hip.free_decode_moments(feats_pad, t, self.decoder.weight.detach(), self.hop_length, ref, mask_pad,
                                       mask_act, out_mode, out)

    # -- reference API -------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[N,L] -> [N,C,T] (encoder.py:71-83)."""
        with torch.no_grad():
            return torch.ops.puresound_amd.free_encode(x, self.encoder.weight, self.hop_length, self.output_active)

    def inverse(self, x: torch.Tensor) -> torch.Tensor:
        """[N,C,T] -> [N,L] (encoder.py:85-94)."""
        with torch.no_grad():
            return torch.ops.puresound_amd.free_decode(x, self.decoder.weight, self.hop_length)


def create_fourier_kernels(n_fft: int):
    """float64 sin/cos tables -> fp32, freq_scale="no" (lobe/stft.py:91-100): [n_fft/2+1, 1, n_fft]."""
    s = torch.arange(n_fft, dtype=torch.float64)
    k = torch.arange(n_fft // 2 + 1, dtype=torch.float64).reshape(-1, 1)
    ang = 2 * math.pi * k * s / n_fft
    return torch.sin(ang).to(torch.float32).unsqueeze(1), torch.cos(ang).to(torch.float32).unsqueeze(1)


def _stft_shape(ctor, x, aux, params):
    n_fft = ctor["n_fft"]
    hop = ctor["hop_length"] if ctor.get("hop_length") else (ctor.get("win_length") or n_fft) // 4
    bins = ctor["freq_bins"] if ctor.get("freq_bins") else n_fft // 2 + 1
    return (x[0], bins, (x[-1] - n_fft) // hop + 1, 2)


def _istft_shape(ctor, x, aux, params):
    n_fft = ctor["n_fft"]
    hop = ctor["hop_length"] if ctor.get("hop_length") else (ctor.get("win_length") or n_fft) // 4
    return (x[0], (x[2] - 1) * hop + n_fft)


def _stft_rebuild(a):
    return dict(a, window_mask=torch.hann_window(a.get("win_length") or a["n_fft"]))


@op_module("istft_decode", _istft_shape, method="_istft", rebuild=_stft_rebuild, cpu="istft_decode")
@op_module("stft_encode", _stft_shape, rebuild=_stft_rebuild, cpu="stft_encode")
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

    def _op_ctor_args(self):
        """JSON-able constructor arguments (the window itself travels as the buffer `window_mask`)."""
        return dict(n_fft=self.n_fft, win_length=self.win_length, freq_bins=self.freq_bins, hop_length=self.stride,
                    iSTFT=self.iSTFT, trainable=self.trainable, output_format=self.output_format)

    # -- kernel-side plans (rebuilt when a table changes) -----------------------------------------------
    def _sig(self):
        return tuple((t.data_ptr(), t._version) for t in list(self.parameters()) + list(self.buffers()))

    def _analysis_plan(self, drop_first_bin: bool):
        """Packed weight of the analysis product in the wrapper's channel order [re bins ; im bins]:
        re = conv(x, wcos), im = -conv(x, wsin) (encoder.py:370-382), optionally without the DC bin
        (base_nn.py:341-345)."""
        key = ("a", drop_first_bin, self._sig())
        if getattr(self, "_plan_a", None) is None or self._plan_a[0] != key:
            lo = 1 if drop_first_bin else 0
            bins = self.freq_bins if self.freq_bins is not None else self.wcos.shape[0]
            w = torch.cat([self.wcos.detach()[lo:bins, 0, :], -self.wsin.detach()[lo:bins, 0, :]], dim=0).float()
            self._plan_a = (key, hip.pack_wt(w), w.shape[0])
        return self._plan_a[1], self._plan_a[2]

    def _synthesis_plan(self, drop_first_bin: bool):
        """Packed weight of the synthesis product with the Hermitian extension (extend_fbins, stft.py:118-125)
        folded in: out[s] = sum_f re[f]*(Kc[s][f] + Kc[s][N-f]) - im[f]*(Ks[s][f] - Ks[s][N-f]), the mirrored
        terms only for 0 < f < N/2 (encoder.py:419-433); input channels [re bins ; im bins]."""
        key = ("s", drop_first_bin, self._sig())
        if getattr(self, "_plan_s", None) is None or self._plan_s[0] != key:
            kc = self.kernel_cos_inv.detach()[:, 0, :, 0].float()  # [out sample s][bin h], h = 0..N-1
            ks = self.kernel_sin_inv.detach()[:, 0, :, 0].float()
            n = self.n_fft
            f = torch.arange(n // 2 + 1, device=kc.device)
            mirror = ((f > 0) & (f < n // 2)).float().unsqueeze(0)
            mf = (n - f) % n
            wre = kc[:, f] + mirror * kc[:, mf]
            wim = -(ks[:, f] - mirror * ks[:, mf])
            lo = 1 if drop_first_bin else 0
            w = torch.cat([wre[:, lo:], wim[:, lo:]], dim=1).contiguous()  # [n_fft, 2*bins]
            self._plan_s = (key, hip.pack_wt(w), self.window_mask.detach().flatten().float().contiguous())
        return self._plan_s[1], self._plan_s[2]

    def __getstate__(self):
        state = dict(self.__dict__)
        state.pop("_plan_a", None)
        state.pop("_plan_s", None)
        return state

    # -- padded-layout entry points used by the fused wrapper ------------------------------------------
    def encode_padded(self, x: torch.Tensor, drop_first_bin: bool):
        """[N,L] -> ([re;im] channel layout, padded [N,2*bins,ldt], T)."""
        hip.require_device(x, "ConvSTFT.forward")
        frames, t = hip.frame(x, self.n_fft, self.stride)
        wt, m = self._analysis_plan(drop_first_bin)
        y, _ = hip.conv1x1(frames, t, wt, m, out=torch.empty(x.shape[0], m, frames.shape[-1], device=x.device))
        return y, t

    def decode_padded(self, spec_pad: torch.Tensor, t: int, drop_first_bin: bool, out_mode: str = "none"):
        """[re;im] channel layout padded [N,2*bins,ldt] -> waveform [N,(T-1)*hop+n_fft]."""
        if not hasattr(self, "kernel_sin_inv") or not hasattr(self, "kernel_cos_inv"):
            raise NameError("Please activate the iSTFT module by setting `iSTFT=True` if you want to use `inverse`")
        wt, window = self._synthesis_plan(drop_first_bin)
        frames, _ = hip.conv1x1(spec_pad, t, wt, self.n_fft,
                                out=torch.empty(spec_pad.shape[0], self.n_fft, spec_pad.shape[-1],
                                                device=spec_pad.device))
        return hip.istft_ola(frames, t, window, self.stride, out_mode)

    # -- reference API ----------------------------------------------------------------------------------
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[N,1,L] -> [N,F,T,2] = stack(real, -imag) ("Complex") or stack(mags, phase) ("MagPhase")
        (encoder.py:358-391)."""
        if self.output_format not in ("Complex", "MagPhase"):
            raise NotImplementedError
        y, t = self.encode_padded(x[:, 0, :] if x.dim() == 3 else x, False)
        if self.output_format == "MagPhase":
            y = hip.magphase(y, bool(self.trainable))
        bins = y.shape[1] // 2
        y = hip.unpad_rows(y, t)
        return torch.stack((y[:, :bins], y[:, bins:]), dim=-1)

    def inverse(self, X: torch.Tensor, refresh_win: bool = True) -> torch.Tensor:
        """[N,F,T,2] -> [N,L] (encoder.py:393-456)."""
        if not hasattr(self, "kernel_sin_inv") or not hasattr(self, "kernel_cos_inv"):
            raise NameError("Please activate the iSTFT module by setting `iSTFT=True` if you want to use `inverse`")
        assert X.dim() == 4, "Inverse iSTFT only works for complex number (batch, freq_bins, timesteps, 2)."
        if self.output_format != "Complex":
            raise NotImplementedError("Inverse only support complex input")
        return self._istft(X)

    def _istft(self, X: torch.Tensor) -> torch.Tensor:
        hip.require_device(X, "ConvSTFT.inverse")
        spec = torch.cat((X[..., 0], X[..., 1]), dim=1).contiguous()
        return self.decode_padded(hip.pad_rows(spec), X.shape[2], False)


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


# ------------------------------------------------------------------------------------------------------------
# Mel front end of the speaker branch (encoder.py:186-272, 459-598; stft.py:129-293)
# ------------------------------------------------------------------------------------------------------------
def _hz_to_mel(f):
    """Slaney mel scale: linear (200/3 Hz per mel) below 1 kHz, logarithmic (27 steps per factor 6.4) above."""
    import numpy as np
    f = np.asarray(f, dtype=np.float64)
    lin = f / (200.0 / 3)
    brk_mel, step = 1000.0 / (200.0 / 3), np.log(6.4) / 27.0
    return np.where(f >= 1000.0, brk_mel + np.log(np.maximum(f, 1e-30) / 1000.0) / step, lin)


def _mel_to_hz(m):
    import numpy as np
    m = np.asarray(m, dtype=np.float64)
    brk_mel, step = 1000.0 / (200.0 / 3), np.log(6.4) / 27.0
    return np.where(m >= brk_mel, 1000.0 * np.exp(step * (m - brk_mel)), (200.0 / 3) * m)


def mel_filterbank(sr: int, n_fft: int, n_banks: int = 128, fmin: float = 0.0, fmax: Optional[float] = None,
                   norm: int = 1) -> torch.Tensor:
    """[n_banks, n_fft//2+1] triangular Slaney-normalised mel weights (same construction as stft.py:237-293)."""
    import numpy as np
    if fmax is None:
        fmax = float(sr / 2)
    bins = np.linspace(0, float(sr) / 2, int(1 + n_fft // 2), endpoint=True)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(fmin), _hz_to_mel(fmax), n_banks + 2))
    width = np.diff(edges)
    ramps = np.subtract.outer(edges, bins)
    w = np.zeros((n_banks, bins.shape[0]), dtype=np.float32)
    for i in range(n_banks):
        w[i] = np.maximum(0, np.minimum(-ramps[i] / width[i], ramps[i + 2] / width[i + 1]))
    if norm == 1:
        w *= (2.0 / (edges[2:n_banks + 2] - edges[:n_banks]))[:, None]
    if not np.all((edges[:-2] == 0) | (w.max(axis=1) > 0)):
        raise ValueError("Empty filters detected in mel frequency basis.")
    return torch.from_numpy(w)


class ConvMelSpectrogram(ConvSTFT):
    """Mel spectrogram on the conv-STFT (encoder.py:459-598): analysis GEMM, power / magnitude of the bins
    (ps_magnitude_f32), mel projection GEMM.  The inverse (pseudo-inverse mel + iSTFT) is not on the HIP path."""

    def __init__(self, window_mask: torch.Tensor, n_fft: int = 512, win_length: int = 512,
                 freq_bins: Optional[int] = None, hop_length: Optional[int] = None, freq_scale: str = "no",
                 iSTFT: bool = True, fmin: int = 50, fmax: int = 6000, sr: int = 16000, trainable: bool = False,
                 output_format: str = "MagPhase", n_banks: int = 80):
        super().__init__(window_mask, n_fft, win_length, freq_bins, hop_length, freq_scale, iSTFT, fmin, fmax, sr,
                         trainable, output_format)
        mel_fb = mel_filterbank(sr=16000, n_fft=n_fft, n_banks=n_banks).permute(1, 0)     # [bins, n_mels]
        inv_mel_fb = torch.pinverse(mel_fb)
        if trainable:
            self.register_parameter("filterbank", nn.Parameter(mel_fb, requires_grad=True))
            self.register_parameter("inv_filterbank", nn.Parameter(inv_mel_fb, requires_grad=True))
        else:
            self.register_buffer("filterbank", mel_fb)
            self.register_buffer("inv_filterbank", inv_mel_fb)

    def encode_padded(self, x: torch.Tensor):
        """[N, L] -> (mel features padded [N, n_mels, ldt], T)."""
        fmt = self.output_format.lower()
        if fmt not in ("magnitude", "magphase"):
            raise NotImplementedError
        if fmt == "magphase":
            raise NotImplementedError("ConvMelSpectrogram on HIP: the 'Magnitude' output (mel of the power spectrum)")
        spec, t = ConvSTFT.encode_padded(self, x, False)                                    # [re bins ; im bins]
        bins = spec.shape[1] // 2
        power = hip.magnitude(spec, t, False, False, kind="power_eps" if self.trainable else "power")
        key = (self.filterbank.data_ptr(), self.filterbank._version)
        if getattr(self, "_mel_plan", None) is None or self._mel_plan[0] != key:
            self._mel_plan = (key, hip.pack_wt(self.filterbank.detach().float().t().contiguous()))
        n_mels = self.filterbank.shape[1]
        y, _ = hip.conv1x1(power, t, self._mel_plan[1], n_mels,
                           out=torch.empty(x.shape[0], n_mels, spec.shape[-1], dtype=torch.float32, device=x.device))
        assert bins == self.filterbank.shape[0]
        return y, t

    def forward(self, x: torch.Tensor):
        """x [N, 1, L] -> mel [N, n_mels, T] (output_format 'Magnitude', encoder.py:529-536)."""
        hip.require_device(x, "ConvMelSpectrogram.forward")
        y, t = self.encode_padded(x[:, 0] if x.dim() == 3 else x)
        return hip.unpad_rows(y, t)

    def inverse(self, melspec, phase, refresh_win=True):
        raise NotImplementedError("ConvMelSpectrogram.inverse (pseudo-inverse mel + iSTFT) is not on the HIP path")


class FbankEnc(nn.Module):
    """Mel front end (encoder.py:186-272); constructor order as the reference."""

    def __init__(self, fft_length: int = 512, win_type: str = "hann", win_length: int = 512, freq_bins: int = None,
                 hop_length: int = 128, freq_scale: str = "no", fmin: int = 0, fmax: int = 8000, sr: int = 16000,
                 trainable: bool = True, output_format: str = "Magnitude", n_banks=80):
        super().__init__()
        self.n_fft, self.win_length, self.freq_bins, self.hop_length = fft_length, win_length, freq_bins, hop_length
        self.freq_scale, self.iSTFT, self.fmin, self.fmax, self.sr = freq_scale, False, fmin, fmax, sr
        self.trainable, self.output_format, self.n_banks = trainable, output_format, n_banks
        if win_type.lower() != "hann":
            raise NotImplementedError("window type not support")
        self.window = torch.hann_window(self.win_length)
        self.encoder = ConvMelSpectrogram(self.window, n_fft=self.n_fft, win_length=self.win_length,
                                          freq_scale=self.freq_scale, iSTFT=self.iSTFT, sr=self.sr, fmin=self.fmin,
                                          fmax=self.fmax, output_format=self.output_format, trainable=self.trainable,
                                          hop_length=self.hop_length, n_banks=self.n_banks)

    def encode_padded(self, x: torch.Tensor):
        return self.encoder.encode_padded(x)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        """[N, L] -> [N, n_banks, T]."""
        return self.encoder(x.unsqueeze(1))

    def inverse(self, magphase: torch.Tensor) -> torch.Tensor:
        return self.encoder.inverse(magphase[..., 0], magphase[..., 1])
