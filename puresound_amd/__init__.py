"""puresound_amd -- MI355X (gfx950) implementation of PureSound's separator forward path.

The package mirrors the part of `puresound.nnet` that lies on the inference path
(encoder -> Conv-TasNet masker -> mask -> decoder -> clamp): same class names, constructor and call
signatures, state_dict keys and error types, with the arithmetic done by hand-written HIP kernels in
libpuresound_hip.so (C ABI: include/puresound_hip.h).  There is no CPU fallback.
"""
__version__ = "0.1.0"
