"""puresound_amd -- MI355X (gfx950) implementation of PureSound's separator forward path.

The package mirrors the part of `puresound.nnet` that lies on the inference path
(encoder -> Conv-TasNet masker -> mask -> decoder -> clamp): same class names, constructor and call
signatures, state_dict keys and error types, with the arithmetic done by hand-written HIP kernels in
libpuresound_hip.so (C ABI: include/puresound_hip.h).  ROCm tensors never fall back to anything else; CPU tensors are
served by stock-ATen registrations of the filterbank / STFT / Conv-TasNet operators (nnet/cpu_path.py), nothing more.
"""
__version__ = "0.1.0"


def _register_operators():
    """torch.ops.puresound_amd.* exist as soon as the package is imported (the module files register them)."""
    from . import nnet  # noqa: F401
    from .streaming import skim_inference  # noqa: F401


_register_operators()
