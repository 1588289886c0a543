"""CPU implementations of the path's operators: stock ATen compositions, used when the tensors are CPU tensors.

BASELINE configs[0] is "PyTorch CPU forward (plumbing, no GPU)" and the recipes' `--backend cpu`
(egs/ns/main.py:184-198, egs/tse/main.py:449-466 of the reference) call `model.inference` on CPU tensors; with this file
the mirror modules serve those calls too.  Nothing here is the product's hot path (that is the HIP library, which still
fails loudly when it is missing on a ROCm device) and nothing here comes from `oracle/`: the oracle is test
infrastructure and spells the arithmetic out with explicit taps and matmuls, this is plain `torch.nn.functional`.

Covered: FreeEncDec, ConvSTFT / ConvEncDec ("Complex"), the norms, DepthwiseSeparableConv1d, TCN, GatedTCN, ConvTasNet
and `SoTaskWrapModule.inference` without a speaker branch.  The recurrent maskers, the U-Net family and the speaker
nets have no CPU path; their operators keep raising on CPU tensors.
"""
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F


# ---- filterbanks -----------------------------------------------------------------------------------------------------
def free_encode(wav: torch.Tensor, weight: torch.Tensor, hop: int, relu: bool) -> torch.Tensor:
    """[N, L] -> [N, C, T]: Conv1d(1 -> C, win, stride hop) [+ ReLU] (lobe/encoder.py:71-83 of the reference)."""
    y = F.conv1d(wav.unsqueeze(1), weight, stride=hop)
    return torch.relu(y) if relu else y


def free_decode(feats: torch.Tensor, weight: torch.Tensor, hop: int) -> torch.Tensor:
    """[N, C, T] -> [N, (T - 1) hop + win]: ConvTranspose1d(C -> 1) (lobe/encoder.py:85-94)."""
    return F.conv_transpose1d(feats, weight, stride=hop).squeeze(1)


def stft_encode(m, x: torch.Tensor) -> torch.Tensor:
    """ConvSTFT.forward: [N, 1, L] or [N, L] -> [N, F, T, 2] = (real, -imag) or (magnitude, phase) (lobe/encoder.py:358-391)."""
    if m.output_format not in ("Complex", "MagPhase"):
        raise NotImplementedError
    x = x if x.dim() == 3 else x.unsqueeze(1)
    real = F.conv1d(x, m.wcos, stride=m.stride)
    imag = F.conv1d(x, m.wsin, stride=m.stride)
    if m.output_format == "Complex":
        return torch.stack((real, -imag), dim=-1)
    mags = torch.sqrt(real.pow(2) + imag.pow(2) + (1e-8 if m.trainable else 0.0))
    return torch.stack((mags, torch.atan2(-imag + 0.0, real)), dim=-1)


def istft_decode(m, X: torch.Tensor) -> torch.Tensor:
    """ConvSTFT.inverse: [N, F, T, 2] -> [N, (T - 1) hop + n_fft] (lobe/encoder.py:393-456): Hermitian extension, the two
    synthesis products, window, overlap-add and the division by the window's overlap-added square where it is > 1e-10."""
    n_fft, hop = m.n_fft, m.stride
    ext = torch.cat((X, X[:, 1:-1].flip(1) * X.new_tensor([1.0, -1.0])), dim=1)   # conj of bins F-2 .. 1
    re, im = ext[..., 0].unsqueeze(1), ext[..., 1].unsqueeze(1)                    # [N, 1, n_fft, T]
    a1 = F.conv2d(re, m.kernel_cos_inv)                                            # kernels [n_fft, 1, n_fft, 1]
    b2 = F.conv2d(im, m.kernel_sin_inv)
    frames = ((a1 - b2) / n_fft).squeeze(2) * m.window_mask                        # [N, n_fft, T]
    t = frames.shape[-1]
    length = (t - 1) * hop + n_fft
    fold = lambda v: F.fold(v, (1, length), (1, n_fft), stride=(1, hop)).reshape(v.shape[0], length)
    wave = fold(frames)
    wsq = fold((m.window_mask ** 2).expand(1, n_fft, t).contiguous())[0]
    return torch.where(wsq > 1e-10, wave / wsq.clamp_min(1e-30), wave)


# ---- norms and the TCN family ----------------------------------------------------------------------------------------
def norm(mod: nn.Module, x: torch.Tensor) -> torch.Tensor:
    from .lobe.norm import ChanLN, GlobLN, InstantLN
    if isinstance(mod, GlobLN):
        dims = tuple(range(1, x.dim()))
        mean = x.mean(dims, keepdim=True)
        var = (x - mean).pow(2).mean(dims, keepdim=True)
        shape = (1, -1) + (1,) * (x.dim() - 2)
        return (x - mean) / torch.sqrt(var + mod.eps) * mod.gamma.reshape(shape) + mod.beta.reshape(shape)
    if isinstance(mod, (ChanLN, InstantLN)):
        if isinstance(mod, InstantLN):
            n, ch, c, t = x.shape
            return norm_chan(mod, x.reshape(n, ch * c, t)).reshape(n, ch, c, t)
        return norm_chan(mod, x)
    if isinstance(mod, nn.GroupNorm):
        return F.group_norm(x, mod.num_groups, mod.weight, mod.bias, mod.eps)
    if isinstance(mod, (nn.BatchNorm1d, nn.BatchNorm2d)):
        return F.batch_norm(x, mod.running_mean, mod.running_var, mod.weight, mod.bias, False, 0.0, mod.eps)
    if isinstance(mod, nn.LayerNorm):
        return F.layer_norm(x, mod.normalized_shape, mod.weight, mod.bias, mod.eps)
    raise NotImplementedError(f"no CPU path for the norm {type(mod).__name__}")


def norm_chan(mod, x: torch.Tensor) -> torch.Tensor:
    mean = x.mean(1, keepdim=True)
    var = x.var(1, keepdim=True, unbiased=False)
    shape = (1, -1) + (1,) * (x.dim() - 2)
    return (x - mean) / torch.sqrt(var + mod.eps) * mod.gamma.reshape(shape) + mod.beta.reshape(shape)


def _conv_norm_act(seq: nn.Sequential, x: torch.Tensor) -> torch.Tensor:
    """Sequential(Conv1d, norm, PReLU[, Dropout, Sigmoid]) of the TCN family, eval mode."""
    conv = seq[0]
    y = F.conv1d(x, conv.weight, conv.bias, conv.stride, conv.padding, conv.dilation, conv.groups)
    y = F.prelu(norm(seq[1], y), seq[2].weight)
    return torch.sigmoid(y) if isinstance(seq[-1], nn.Sigmoid) else y


def depthwise_separable(m, x: torch.Tensor) -> torch.Tensor:
    """DepthwiseSeparableConv1d.forward (lobe/cnn.py:84-106)."""
    h = _conv_norm_act(m.in_conv, x) if m.transform else x
    h = _conv_norm_act(m.depthwise, h)
    h = _conv_norm_act(m.pointwise, h)
    if m.skip:
        h = h + F.conv1d(x, m.skip_conv.weight, m.skip_conv.bias)
    return h[..., :-m.padding] if (m.causal and m.padding) else h


def tcn_block(m, x: torch.Tensor, embed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """TCN.forward (conv_tasnet.py:67-90)."""
    if (embed is not None) != (m.emb_dim > 0):
        raise RuntimeError(f"TCN.forward: block built with emb_dim={m.emb_dim} but embed is "
                           f"{'given' if embed is not None else 'missing'} (the reference fails in in_conv)")
    h = x if embed is None else torch.cat((x, embed.unsqueeze(2).expand(-1, -1, x.shape[-1])), dim=1)
    h = _conv_norm_act(m.in_conv, h)
    h = depthwise_separable(m.dconv[0], h)
    return F.conv1d(h, m.out_conv.weight, m.out_conv.bias) + x


def gated_tcn_block(m, x: torch.Tensor, embed: Optional[torch.Tensor] = None) -> torch.Tensor:
    """GatedTCN.forward (conv_tasnet.py:178-215)."""
    h = F.conv1d(x, m.in_conv.weight)
    left = _conv_norm_act(m.left_conv, h)
    r = h
    if embed is not None:
        e = embed.unsqueeze(2)
        if m.use_film:
            r = F.conv1d(e, m.cond_scale.weight) * h + F.conv1d(e, m.cond_bias.weight)
        else:
            r = torch.cat((h, e.expand(-1, -1, h.shape[-1])), dim=1)
    right = _conv_norm_act(m.right_conv, r)
    y = F.conv1d(left * right, m.out_conv.weight)
    if m.causal and m.padd:
        y = y[..., :-m.padd]
    return y + x


def conv_tasnet(m, x: torch.Tensor, dvec: Optional[torch.Tensor] = None) -> torch.Tensor:
    """ConvTasNet.forward (conv_tasnet.py:338-359)."""
    from .conv_tasnet import TCN
    if dvec is not None and m.embed_norm:
        dvec = F.normalize(dvec, p=2, dim=1)
    for stack in m.tcn_list:
        for i, blk in enumerate(stack):
            e = dvec if (m.tcn_with_embed[i] and dvec is not None) else None
            x = tcn_block(blk, x, e) if isinstance(blk, TCN) else gated_tcn_block(blk, x, e)
    return x


# ---- the wrapper -----------------------------------------------------------------------------------------------------
def wrapper_inference(w, noisy: torch.Tensor, enroll: Optional[torch.Tensor] = None) -> torch.Tensor:
    """SoTaskWrapModule.inference on CPU tensors (base_nn.py:690-722): _get_feature -> masker -> get_mask ->
    apply_tf_masks -> _get_waveform -> _wav_output_constrain, through the modules' reference-API calls."""
    from .conv_tasnet import ConvTasNet
    from .lobe.encoder import ConvEncDec, FreeEncDec
    if enroll is not None or w.embedding_free_tse:
        raise NotImplementedError("CPU path: inference without a speaker branch (the speaker nets run on the HIP path only)")
    if not isinstance(w.masker, ConvTasNet) or not isinstance(w.encoder, (FreeEncDec, ConvEncDec)):
        raise NotImplementedError("CPU path: FreeEncDec / ConvEncDec encoder with a ConvTasNet masker")
    mask_act = w.check_mask_constraint(w.mask_constraint)
    pairing = w.check_mask_pairing(w.mask_type, w.f_type)
    out_mode = w.output_constraint.lower()
    if out_mode not in ("linear", "sigmoid"):
        raise NameError("Non support type.")  # base_nn.py:421-422
    with torch.no_grad():
        stft = isinstance(w.encoder, ConvEncDec)
        feats = w.encoder(noisy)
        if stft:                                                 # base_nn.py:337-345
            lo = 1 if w.drop_first_bin else 0
            feats = torch.cat((feats[:, lo:, :, 0], feats[:, lo:, :, 1]), dim=1)
        mask = w.masker(feats)
        mask = {"linear": lambda v: v, "relu": torch.relu, "sigmoid": torch.sigmoid}[mask_act](mask)
        if pairing == "real":
            enh = feats * mask
        elif pairing == "complex":                               # base_nn.py:56-61, 97-112
            fr, fi = feats.chunk(2, dim=1)
            mr, mi = mask.chunk(2, dim=1)
            enh = torch.cat((fr * mr - fi * mi, fr * mi + fi * mr), dim=1)
        else:
            raise NotImplementedError("CPU path: (real, real) or (complex, complex) masks")
        if stft:                                                 # base_nn.py:380-395
            re, im = enh.chunk(2, dim=1)
            if w.drop_first_bin:
                re, im = F.pad(re, (0, 0, 1, 0)), F.pad(im, (0, 0, 1, 0))
            wav = w.encoder.inverse(torch.stack((re, im), dim=-1))
        else:
            wav = w.encoder.inverse(enh)
        return wav.clamp_(-1.0, 1.0) if out_mode == "linear" else torch.sigmoid(wav)
