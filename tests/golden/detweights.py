"""Deterministic, name-keyed weights for parity fixtures.

The same function fills a reference model (inside make_golden.py, in the build
container) and this repo's modules / the oracle (in tests, here and on the GPU
box), so no reference file has to travel.  Values depend only on the state_dict
key and the tensor shape.
"""
import zlib

import numpy as np
import torch


def _gen(name: str):
    return np.random.Generator(np.random.Philox(key=zlib.crc32(name.encode())))


def det_tensor(name: str, ref: torch.Tensor, mode: str = "plain") -> torch.Tensor:
    """mode "wild": what a trained checkpoint may look like and formula weights do not -- norm gains log-uniform over
    2^-4 .. 2^4, norm / conv biases in +-2, PReLU slopes up to 3 (beyond 1 a negative value GROWS in the activation), and
    in every weight matrix a handful of rows scaled so that the rows span 2^18 in magnitude."""
    shape = tuple(ref.shape)
    g = _gen(name)
    leaf = name.rsplit(".", 1)[-1]
    # (the encoder / decoder filterbanks keep the plain law: with 2^9 rows in them every output sample saturates the clamp)
    wild = mode == "wild" and name.startswith(("masker.", "speaker_net."))
    if wild and leaf not in ("num_batches_tracked", "running_mean", "running_var"):
        if leaf in ("gamma",) or (leaf == "weight" and len(shape) == 1 and shape[0] > 1):
            v = np.exp2(g.uniform(-4.0, 4.0, shape))
        elif leaf in ("beta",) or (leaf == "bias" and len(shape) == 1):
            v = g.uniform(-2.0, 2.0, shape)
        elif leaf == "weight" and shape == (1,):
            v = g.uniform(0.05, 3.0, shape)
        else:
            fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
            b = 1.0 / np.sqrt(max(fan_in, 1))
            v = g.uniform(-b, b, shape)
            if len(shape) > 1 and shape[0] >= 16:
                rows = g.choice(shape[0], size=4, replace=False)
                for r, e in zip(rows, (9.0, -9.0, 6.0, -7.0)):
                    v[r] *= 2.0 ** e
        return torch.tensor(np.asarray(v), dtype=ref.dtype).reshape(shape)
    if leaf == "num_batches_tracked":
        return torch.zeros(shape, dtype=ref.dtype)
    if leaf == "running_mean":
        v = g.uniform(-0.2, 0.2, shape)
    elif leaf == "running_var":
        v = g.uniform(0.5, 1.5, shape)
    elif leaf in ("gamma",) or (leaf == "weight" and len(shape) == 1 and shape[0] > 1):
        v = g.uniform(0.5, 1.5, shape)  # norm gains
    elif leaf in ("beta",) or (leaf == "bias" and len(shape) == 1):
        v = g.uniform(-0.2, 0.2, shape)  # norm / conv biases
    elif leaf == "weight" and shape == (1,):
        v = g.uniform(0.1, 0.4, shape)  # PReLU slope
    else:
        fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
        b = 1.0 / np.sqrt(max(fan_in, 1))
        v = g.uniform(-b, b, shape)
    return torch.tensor(np.asarray(v), dtype=ref.dtype).reshape(shape)


# buffers that are constants of the algorithm, never randomised
_KEEP = ("kernel_sin_inv", "kernel_cos_inv", "window_mask", "pe")


def det_state_dict(model: torch.nn.Module, perturb_stft: float = 0.02, mode: str = "plain") -> dict:
    """New state_dict for `model`, same keys/shapes.  STFT analysis kernels (wsin/wcos) keep their
    Fourier initialisation plus a small name-keyed perturbation (they are trainable in every recipe,
    so a pure-FFT shortcut must not pass)."""
    out = {}
    for k, v in model.state_dict().items():
        leaf = k.rsplit(".", 1)[-1]
        if leaf in _KEEP:
            out[k] = v.clone()
        elif leaf in ("wsin", "wcos"):
            g = _gen(k)
            out[k] = v.clone() + torch.tensor(g.uniform(-perturb_stft, perturb_stft, tuple(v.shape)), dtype=v.dtype)
        else:
            out[k] = det_tensor(k, v, mode)
    return out


def det_wave(seed: int, n: int, length: int, amp: float = 0.5) -> torch.Tensor:
    g = np.random.Generator(np.random.Philox(key=seed))
    return torch.tensor(g.uniform(-amp, amp, (n, length)), dtype=torch.float32)
