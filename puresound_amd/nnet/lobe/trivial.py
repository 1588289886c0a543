"""Conditioning lobes of the recurrent maskers (mirror of puresound/nnet/lobe/trivial.py:61-167).

FiLM: x' = LN(x); [scale; bias] = one stacked 1x1 conv on [x'; e] (the embedding columns become a per-utterance
bias, ps_embed_bias_f32); out = scale * x' + bias.  Gate: in_conv -> (ChanLN+PReLU) * sigmoid(ChanLN+PReLU) ->
out_conv + x.  Both work on the padded channel-major layout; the reference's [N,C,T] forward() is kept.
"""
from typing import Optional

import torch
import torch.nn as nn

from ... import hip
from ...ops import op_module
from .._plans import PlanCache, _f32, layernorm_plan
from .norm import ChanLN


def _magnitude_shape(ctor, x, aux, params):
    drop = int(bool(ctor.get("drop_first", True)))
    if len(x) == 4:  # [N, H, T, 2]
        return (x[0], x[1] - drop, x[2])
    return (x[0], x[1] // 2 - drop, x[2])


@op_module("magnitude_fwd", _magnitude_shape)
class Magnitude(nn.Module):
    """STFT [re; im] -> magnitude (lobe/trivial.py:21-59): 3-D input [N, 2H, T] (channel halves, what the wrapper
    hands the speaker net) or the encoder's own 4-D [N, H, T, 2]."""

    def __init__(self, drop_first: bool = True, log1p: bool = False) -> None:
        super().__init__()
        self.drop_first = drop_first
        self.log1p = log1p

    def forward_padded(self, x: torch.Tensor, t: int) -> torch.Tensor:
        if x.shape[1] % 2:
            raise TypeError
        return hip.magnitude(x, t, self.drop_first, self.log1p)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        hip.require_device(x, "Magnitude.forward")
        if x.dim() == 4:  # [N, H, T, 2] -> the channel-halves form (trivial.py:40-42)
            if x.shape[-1] != 2:
                raise TypeError
            x = torch.cat([x[..., 0], x[..., 1]], dim=1)
        elif x.dim() != 3:
            raise TypeError
        t = x.shape[-1]
        return hip.unpad_rows(self.forward_padded(hip.pad_rows(x), t), t)


class SpecAugment(nn.Module):
    """Random frequency / time masking of a spectrogram [N, F, T] (lobe/trivial.py:306-335).  The reference applies it
    in eval mode too (its forward has no training switch), through torchaudio.functional.mask_along_axis; that function's
    algorithm is restated here draw for draw -- per masked axis `value = torch.rand(1) * mask_param`,
    `min_value = torch.rand(1) * (size - value)`, span [long(min_value), long(min_value) + long(value)), one span for the
    whole batch, both draws from torch's global CPU generator -- so a seeded run masks what the reference masks.  The
    fill itself is ps_fill_span_f32."""

    def __init__(self, freq_mask_length: int, time_mask_length: int, fill_value: float) -> None:
        super().__init__()
        self.freq_mask = freq_mask_length
        self.time_mask = time_mask_length
        self.mask_value = fill_value

    @staticmethod
    def _span(size: int, mask_param: int):
        value = torch.rand(1) * mask_param
        min_value = torch.rand(1) * (size - value)
        lo = int(min_value.long())
        return lo, lo + int(value.long())

    def forward_padded(self, x: torch.Tensor, t: int) -> torch.Tensor:
        """[N, F, ld] rows with t valid frames."""
        if self.freq_mask != 0:
            lo, hi = self._span(x.shape[1], self.freq_mask)
            x = hip.fill_span(x, 1, lo, hi, self.mask_value)
        if self.time_mask != 0:
            lo, hi = self._span(t, self.time_mask)
            x = hip.fill_span(x, 2, lo, hi, self.mask_value)
        return x

    def apply_mask(self, x: torch.Tensor) -> torch.Tensor:
        hip.require_device(x, "SpecAugment.forward")
        if x.dim() != 3:
            raise ValueError("SpecAugment: [N, F, T] input")
        t = x.shape[-1]
        return hip.unpad_rows(self.forward_padded(hip.pad_rows(x.float()), t), t)

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return self.apply_mask(x)


class _PerFrameCondition:
    """Streaming: the frames of the row are concurrent streams, each with its own embedding, so the embedding
    columns of the conditioning conv become a per-frame additive term [1, M, ldB] instead of a per-utterance bias.
    The term lives in one buffer that is refreshed in place when the embeddings change, so a captured hipGraph
    keeps reading the right memory."""

    _per_frame = None

    _per_frame_pairs = None

    def set_per_frame_condition(self, condition: torch.Tensor, normalize: bool) -> None:
        p = self._plan_get(condition.device, self._build)
        for attr, key in (("_per_frame", "w_embed"), ("_per_frame_pairs", "w_embed_pairs")):
            if key not in p:
                continue
            rows = hip.embed_bias(condition.float(), p[key], normalize)             # [B, M]
            term = hip.pad_rows(rows.t().unsqueeze(0))                               # [1, M, ldB]
            cur = getattr(self, attr)
            if cur is None or cur.shape != term.shape or cur.device != term.device:
                setattr(self, attr, term)
            else:
                cur.copy_(term)

    def _embed_term(self, condition: torch.Tensor, w_embed: torch.Tensor, normalize: bool, per_frame: bool):
        """-> (bias_n, res) for ps_conv1x1_f32."""
        if not per_frame:
            return hip.embed_bias(condition.float(), w_embed, normalize), None
        if self._per_frame is None:
            raise RuntimeError("per-frame conditioning was not initialised (set_per_frame_condition)")
        return None, self._per_frame


class FiLM(_PerFrameCondition, PlanCache, nn.Module):
    """Feature-wise linear modulation (lobe/trivial.py:129-167)."""

    def __init__(self, feats_size: int, embed_size: int, input_norm: bool = True):
        super().__init__()
        self.feats_size, self.embed_size = feats_size, embed_size
        self.cond_scale = nn.Conv1d(feats_size + embed_size, feats_size, kernel_size=1, bias=False)
        self.cond_bias = nn.Conv1d(feats_size + embed_size, feats_size, kernel_size=1, bias=False)
        self.inp_norm = input_norm
        if self.inp_norm:
            self.norm = nn.LayerNorm(feats_size)

    def _build(self, device):
        c = self.feats_size
        ws, wb = _f32(self.cond_scale.weight, device)[:, :, 0], _f32(self.cond_bias.weight, device)[:, :, 0]
        p = dict(wt=hip.pack_wt(torch.cat([ws[:, :c], wb[:, :c]], 0)),
                 w_embed=torch.cat([ws[:, c:], wb[:, c:]], 0).contiguous())
        # rows paired (scale c, bias c) for the one-kernel streaming form (ps_film_conv_f32)
        pairs = torch.stack([ws, wb], 1).reshape(2 * c, -1)
        p["wt_pairs"] = hip.pack_wt(pairs[:, :c].contiguous())
        p["w_embed_pairs"] = pairs[:, c:].contiguous()
        if self.inp_norm:
            p["norm"] = layernorm_plan(self.norm, device)
        return p

    def forward_padded(self, x: torch.Tensor, t: int, condition: torch.Tensor, normalize: bool = False,
                       per_frame: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x padded [N,C,ldt], condition [N,E] -> padded [N,C,ldt].  `normalize` folds the caller's
        F.normalize(condition) into the embedding GEMV.  per_frame: condition [T,E], one embedding per frame."""
        p = self._plan_get(x.device, self._build)
        n, c, ldt = x.shape
        if self.inp_norm:
            x = hip.chan_layernorm(x, t, p["norm"]["gamma"], p["norm"]["beta"], p["norm"]["eps"])
        bias_n, res = self._embed_term(condition, p["w_embed"], normalize, per_frame)
        sb, _ = hip.conv1x1(x, t, p["wt"], 2 * c, None, None, bias_n, res,
                            out=torch.empty(n, 2 * c, ldt, dtype=torch.float32, device=x.device))
        return hip.film_apply(x, sb, t, out)

    def step_normed(self, xn: torch.Tensor, t: int, out: torch.Tensor) -> torch.Tensor:
        """Streaming: xn is the ALREADY input-normalised frame block [1,C,ldB]; one kernel, per-frame embeddings
        from set_per_frame_condition."""
        p = self._plan_get(xn.device, self._build)
        return hip.film_conv(xn, t, p["wt_pairs"], self._per_frame_pairs, out)

    def forward(self, x: torch.Tensor, condition: torch.Tensor) -> torch.Tensor:
        """x [N,C,T], condition [N,E] -> [N,C,T]."""
        hip.require_device(x, "FiLM.forward")
        t = x.shape[-1]
        return hip.unpad_rows(self.forward_padded(hip.pad_rows(x), t, condition), t)


class Gate(_PerFrameCondition, PlanCache, nn.Module):
    """Gated conditioning (lobe/trivial.py:61-126)."""

    def __init__(self, input_size: int, hidden_size: int, embed_size: int, dropout: float = 0.0):
        super().__init__()
        self.input_size, self.hidden_size, self.embed_size = input_size, hidden_size, embed_size
        self.in_conv = nn.Conv1d(input_size, hidden_size, kernel_size=1, bias=False, groups=1)
        self.left_conv = nn.Sequential(
            nn.Conv1d(hidden_size, hidden_size, kernel_size=1, dilation=1, bias=False, padding=0, groups=1),
            ChanLN(hidden_size), nn.PReLU(), nn.Dropout(p=dropout))
        self.right_conv = nn.Sequential(
            nn.Conv1d(hidden_size + embed_size, hidden_size, kernel_size=1, dilation=1, bias=False, padding=0, groups=1),
            ChanLN(hidden_size), nn.PReLU(), nn.Dropout(p=dropout), nn.Sigmoid())
        self.out_conv = nn.Conv1d(hidden_size, input_size, kernel_size=1, bias=False, groups=1)

    def _build(self, device):
        if self.training and (self.left_conv[3].p > 0 or self.right_conv[3].p > 0):
            raise RuntimeError("Gate: dropout is active; the HIP path is inference only -- call .eval()")
        h = self.hidden_size
        wr = _f32(self.right_conv[0].weight, device)[:, :, 0]
        return dict(w_in=hip.pack_wt(_f32(self.in_conv.weight, device)),
                    w_left=hip.pack_wt(_f32(self.left_conv[0].weight, device)),
                    w_right=hip.pack_wt(wr[:, :h].contiguous()), w_embed=wr[:, h:].contiguous(),
                    w_out=hip.pack_wt(_f32(self.out_conv.weight, device)),
                    ln_left=layernorm_plan(self.left_conv[1], device), ln_right=layernorm_plan(self.right_conv[1], device),
                    slope_left=_f32(self.left_conv[2].weight, device), slope_right=_f32(self.right_conv[2].weight, device))

    def forward_padded(self, x: torch.Tensor, t: int, condition: torch.Tensor, normalize: bool = False,
                       per_frame: bool = False, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        p = self._plan_get(x.device, self._build)
        n, c, ldt = x.shape
        h = self.hidden_size
        new = lambda rows: torch.empty(n, rows, ldt, dtype=torch.float32, device=x.device)  # noqa: E731
        y, _ = hip.conv1x1(x, t, p["w_in"], h, out=new(h))
        a, _ = hip.conv1x1(y, t, p["w_left"], h, out=new(h))
        left = hip.chan_layernorm(a, t, p["ln_left"]["gamma"], p["ln_left"]["beta"], p["ln_left"]["eps"],
                                  slope=p["slope_left"])
        bias_n, res = self._embed_term(condition, p["w_embed"], normalize, per_frame)
        b, _ = hip.conv1x1(y, t, p["w_right"], h, None, None, bias_n, res, out=new(h))
        prod = hip.chan_layernorm(b, t, p["ln_right"]["gamma"], p["ln_right"]["beta"], p["ln_right"]["eps"],
                                  slope=p["slope_right"], sigmoid=True, mul=left)
        y, _ = hip.conv1x1(prod, t, p["w_out"], c, res=x, out=out if out is not None else new(c))
        return y

    def forward(self, x: torch.Tensor, condition: torch.Tensor) -> torch.Tensor:
        hip.require_device(x, "Gate.forward")
        t = x.shape[-1]
        return hip.unpad_rows(self.forward_padded(hip.pad_rows(x), t, condition), t)
