"""PyTorch custom-op boundary: `torch.ops.puresound_amd.*`.

The reference has no FFI; its seam is the `nn.Module` contract, and one caller on the path needs more than eager
calls: the recipe's export action traces `Sequential(encoder, *speaker_net)`, `encoder`, `encoder.decoder` and
`masker` with `torch.jit.trace` (egs/tse/main.py:406-443).  A ctypes launch on `data_ptr()` is invisible to a tracer,
so every reference-API `forward` / `inverse` of the mirror modules goes through an operator registered here with

  * a HIP implementation (dispatch key CUDA) that calls the C ABI of libpuresound_hip.so (through `hip.py` and the
    modules' padded-layout code),
  * a Meta implementation that only computes the output shape (FakeTensor / meta-device tracing, shape checks on a
    box without a GPU),
  * a CPU registration: a stock-ATen composition (nnet/cpu_path.py) for the filterbanks, the STFT pair and the TCN family
    -- what BASELINE's "PyTorch CPU forward" configuration and the recipes' `--backend cpu` need -- and an error for the
    rest.  (On a ROCm device nothing falls back to it: a missing HIP library still fails at first use.)

Two kinds of operators:

  * functional ones whose tensors are all arguments: `free_encode(wav, weight, hop, relu)`,
    `free_decode(feats, weight, hop)`;
  * module-backed ones, `<kind>_fwd(x, aux, params, cfg)`: `params` are the module's parameters and buffers (so a
    traced graph holds them as inputs, not as baked constants), `cfg` the JSON of its constructor arguments.  The
    implementation finds the live module that made the call through a weak registry keyed by (kind, cfg, data
    pointers); when there is none -- a traced module loaded in a fresh process -- it rebuilds the module from `cfg`
    and adopts `params` without copying.

The operators are forward only: the mirror modules call them under `torch.no_grad()` (no autograd formula is registered;
an output never requires grad).

The fused wrapper path (`SoTaskWrapModule.inference`) does not go through these operators: it keeps the padded device
layout from the encoder kernel to the decoder kernel.  The operators are the module-level API.
"""
from __future__ import annotations

import inspect
import json
import weakref
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import torch

NAMESPACE = "puresound_amd"
_LIB = torch.library.Library(NAMESPACE, "DEF")
_KINDS: Dict[str, dict] = {}                       # kind -> {cls, method, shape}
_LIVE: "weakref.WeakValueDictionary" = weakref.WeakValueDictionary()   # (kind, cfg, ptrs) -> module that called
_REBUILT: Dict[tuple, torch.nn.Module] = {}        # modules rebuilt from (cfg, params) of a loaded trace
OP_NAMES: List[str] = []
CPU_OPS: List[str] = []   # operators whose CPU registration computes (stock ATen compositions, nnet/cpu_path.py)


def _no_cpu(name: str) -> Callable:
    def impl(*args, **kwargs):
        raise RuntimeError(f"{NAMESPACE}::{name}: inputs must be HIP (cuda) tensors -- this operator has no CPU path "
                           f"(those that have one: {', '.join(CPU_OPS)})")
    return impl


def _define(name: str, schema: str, hip_impl: Callable, meta_impl: Callable, cpu_impl: Optional[Callable] = None) -> None:
    _LIB.define(f"{name}{schema}")
    _LIB.impl(name, hip_impl, "CUDA")
    _LIB.impl(name, meta_impl, "Meta")
    _LIB.impl(name, cpu_impl if cpu_impl is not None else _no_cpu(name), "CPU")
    OP_NAMES.append(name)
    if cpu_impl is not None:
        CPU_OPS.append(name)


# ---------------------------------------------------------------------------------------------------------------------
# functional operators: learned filterbank (lobe/encoder.py:71-94 of the reference)
# ---------------------------------------------------------------------------------------------------------------------
def _free_encode_hip(wav: torch.Tensor, weight: torch.Tensor, hop: int, relu: bool) -> torch.Tensor:
    from . import hip
    feats, t = hip.free_encode(wav, weight.detach(), hop, relu)
    return hip.unpad_rows(feats, t)


def _free_encode_meta(wav, weight, hop, relu):
    n, length = wav.shape
    c, _, win = weight.shape
    if length < win:
        raise RuntimeError(f"free_encode: input length {length} is shorter than the window {win}")
    return wav.new_empty((n, c, (length - win) // hop + 1))


def _free_decode_hip(feats: torch.Tensor, weight: torch.Tensor, hop: int) -> torch.Tensor:
    from . import hip
    return hip.free_decode(hip.pad_rows(feats), feats.shape[-1], weight.detach(), hop)


def _free_decode_meta(feats, weight, hop):
    n, _, t = feats.shape
    return feats.new_empty((n, (t - 1) * hop + weight.shape[-1]))


def _free_encode_cpu(wav, weight, hop, relu):
    from .nnet import cpu_path
    return cpu_path.free_encode(wav, weight.detach(), hop, relu)


def _free_decode_cpu(feats, weight, hop):
    from .nnet import cpu_path
    return cpu_path.free_decode(feats, weight.detach(), hop)


_define("free_encode", "(Tensor wav, Tensor weight, int hop, bool relu) -> Tensor", _free_encode_hip, _free_encode_meta,
        _free_encode_cpu)
_define("free_decode", "(Tensor feats, Tensor weight, int hop) -> Tensor", _free_decode_hip, _free_decode_meta,
        _free_decode_cpu)


# ---------------------------------------------------------------------------------------------------------------------
# functional operator: an LSTM recurrence over precomputed gate pre-activations (what nn.LSTM does after its input
# projection; the DPRNN / SkiM / SingleRNN paths run it as ps_lstm_f32 behind one GEMM for all frames)
# ---------------------------------------------------------------------------------------------------------------------
def _lstm_seq_hip(gx: torch.Tensor, whh: torch.Tensor, h0: Optional[torch.Tensor], c0: Optional[torch.Tensor]):
    """gx [N, T, 4H] = W_ih x_t + b_ih + b_hh (gate order i, f, g, o as in nn.LSTM.weight_ih_l0), whh [4H, H] =
    weight_hh_l0, optional h0 / c0 [N, H]  ->  (y [N, T, H], h_T [N, H], c_T [N, H])."""
    from . import hip
    n, t, rows = gx.shape
    hid = whh.shape[1]
    if rows != 4 * hid or whh.shape[0] != 4 * hid:
        raise RuntimeError(f"lstm_seq_fwd: gx [N, T, 4H] and whh [4H, H] expected, got {tuple(gx.shape)} / {tuple(whh.shape)}")
    g_rows = hip.pad_rows(gx.float().transpose(1, 2))                     # [N, 4H, ldt]
    whh_t = whh.detach().float().t().contiguous().reshape(1, hid, 4 * hid)
    st = []
    for v in (h0, c0):
        st.append(None if v is None else hip.pad_rows(v.float().reshape(n, hid, 1)))
    if (st[0] is None) != (st[1] is None):   # one of the two given: the other starts at zero
        z = hip.pad_rows(torch.zeros(n, hid, 1, dtype=torch.float32, device=gx.device))
        st = [z if v is None else v for v in st]
    hseq, state = hip.lstm(g_rows, whh_t, hid, 1, 1, g_rows.shape[2], t, 1, st[0], st[1], want_state=True)
    return hip.unpad_rows(hseq, t).transpose(1, 2).contiguous(), state[0][:, :, 0].contiguous(), state[1][:, :, 0].contiguous()


def _lstm_seq_meta(gx, whh, h0, c0):
    n, t, rows = gx.shape
    hid = whh.shape[1]
    if rows != 4 * hid or whh.shape[0] != 4 * hid:
        raise RuntimeError(f"lstm_seq_fwd: gx [N, T, 4H] and whh [4H, H] expected, got {tuple(gx.shape)} / {tuple(whh.shape)}")
    return gx.new_empty((n, t, hid)), gx.new_empty((n, hid)), gx.new_empty((n, hid))


_define("lstm_seq_fwd", "(Tensor gx, Tensor whh, Tensor? h0, Tensor? c0) -> (Tensor, Tensor, Tensor)", _lstm_seq_hip,
        _lstm_seq_meta)


# ---------------------------------------------------------------------------------------------------------------------
# module-backed operators
# ---------------------------------------------------------------------------------------------------------------------
def _call_cache(m: torch.nn.Module, kind: str):
    """(params, cfg) of a module-backed call.  The parameter list is read off the module every time (a load_state_dict
    with assign=True, .to(), a swapped sub-module all change it); the JSON of the constructor arguments -- a constant of
    the module -- is serialised once."""
    cfg = m.__dict__.get("_op_cfg_json")
    if cfg is None:
        cfg = json.dumps({"ctor": getattr(m, "_ctor_args", None)}, sort_keys=True)
        object.__setattr__(m, "_op_cfg_json", cfg)
    params = module_tensors(m)
    if params:
        _LIVE[(kind, cfg, tuple(p.data_ptr() for p in params))] = m
    return params, cfg


def module_tensors(m: torch.nn.Module) -> List[torch.Tensor]:
    """Parameters then buffers, in registration order (the order `adopt_tensors` assigns them back in)."""
    return [p for _, p in m.named_parameters()] + [b for _, b in m.named_buffers()]


def adopt_tensors(m: torch.nn.Module, tensors: Sequence[torch.Tensor]) -> None:
    names = [k for k, _ in m.named_parameters()] + [k for k, _ in m.named_buffers()]
    if len(names) != len(tensors):
        raise RuntimeError(f"{type(m).__name__}: {len(tensors)} tensors for {len(names)} parameters / buffers")
    m.load_state_dict(dict(zip(names, tensors)), strict=False, assign=True)


def _resolve(kind: str, cfg: str, params: Sequence[torch.Tensor]) -> torch.nn.Module:
    key = (kind, cfg, tuple(p.data_ptr() for p in params))
    m = _LIVE.get(key)
    if m is None:
        m = _REBUILT.get(key)
    if m is None:
        info = _KINDS[kind]
        args = json.loads(cfg)
        if args.get("ctor") is None:
            raise RuntimeError(f"{NAMESPACE}::{kind}: the module that recorded this call is gone and its constructor "
                               f"arguments are not serialisable")
        m = info["cls"](**info["rebuild"](args["ctor"])).eval()
        adopt_tensors(m, params)
        while len(_REBUILT) >= 64:   # bounded; the oldest rebuilt module goes first
            _REBUILT.pop(next(iter(_REBUILT)))
        _REBUILT[key] = m
    return m


def op_module(kind: str, shape: Callable, method: str = "forward", rebuild: Optional[Callable] = None,
              cpu: Optional[str] = None):
    """Class decorator: the reference-API `method(x, aux=None)` of the class goes through
    torch.ops.puresound_amd.<kind>; the original body stays reachable as `_hip_<method>`.

    shape(ctor_args, x_shape, aux_shape) -> output shape (the Meta implementation).
    cpu: name of the function in nnet/cpu_path.py that serves CPU tensors, `fn(module, x[, aux])` (None: CPU tensors raise)."""

    def deco(cls):
        body = getattr(cls, method)
        n_in = len([p for p in inspect.signature(body).parameters.values()
                    if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]) - 1
        hip_name = f"_hip_{method}"
        setattr(cls, hip_name, body)
        if "_ctor_sig" not in cls.__dict__:
            init = cls.__init__
            sig = inspect.signature(init)

            def __init__(self, *a, **k):
                init(self, *a, **k)
                try:
                    ba = sig.bind(self, *a, **k)
                    ba.apply_defaults()
                    args = {n: v for n, v in list(ba.arguments.items())[1:]}
                    json.dumps(args)
                except (TypeError, ValueError):
                    args = None
                if hasattr(self, "_op_ctor_args"):
                    args = self._op_ctor_args()
                object.__setattr__(self, "_ctor_args", args)

            __init__.__wrapped__ = init
            cls.__init__ = __init__
            cls._ctor_sig = sig

        def hip_impl(x, aux, params, cfg):
            m = _resolve(kind, cfg, params)
            fn = getattr(m, hip_name)
            return fn(x) if n_in == 1 else fn(x, aux)

        def meta_impl(x, aux, params, cfg):
            out = shape(json.loads(cfg).get("ctor") or {}, tuple(x.shape), None if aux is None else tuple(aux.shape),
                        [tuple(p.shape) for p in params])
            return x.new_empty(out)

        cpu_impl = None
        if cpu is not None:
            def cpu_impl(x, aux, params, cfg):
                from .nnet import cpu_path
                m = _resolve(kind, cfg, params)
                fn = getattr(cpu_path, cpu)
                return fn(m, x) if n_in == 1 else fn(m, x, aux)

        _define(kind, "(Tensor x, Tensor? aux, Tensor[] params, str cfg) -> Tensor", hip_impl, meta_impl, cpu_impl)
        _KINDS[kind] = dict(cls=cls, method=method, shape=shape, rebuild=rebuild or (lambda a: a))
        op = getattr(getattr(torch.ops, NAMESPACE), kind)

        names = [p.name for p in inspect.signature(body).parameters.values()
                 if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)][1:]   # e.g. ["x", "embed"] / ["x", "dvec"]

        def routed(self, *args, **kw):
            # the reference's own call forms keep working: forward(x), forward(x, e), forward(x, embed=e), forward(x=..)
            if len(args) > len(names) or any(k not in names for k in kw) or any(n in kw for n in names[:len(args)]):
                raise TypeError(f"{cls.__name__}.{method}() takes {names}, got {len(args)} positional and {sorted(kw)}")
            vals = dict(zip(names, args))
            vals.update(kw)
            if names[0] not in vals:
                raise TypeError(f"{cls.__name__}.{method}() is missing {names[0]!r}")
            x = vals[names[0]]
            aux = vals.get(names[1]) if len(names) > 1 else None
            # forward only: no backward kernel is registered, so a caller who expects gradients is told, not handed a
            # detached tensor (SoTaskWrapModule's training forwards raise for the same reason)
            if torch.is_grad_enabled() and self.training and (x.requires_grad or any(p.requires_grad for p in self.parameters())):
                raise RuntimeError(f"{cls.__name__}.{method}: the HIP path is inference only (no backward kernels) -- "
                                   f"call .eval() or run under torch.no_grad()")
            params, cfg = _call_cache(self, kind)
            with torch.no_grad():  # outputs carry no graph
                return op(x, aux, params, cfg)

        routed.__doc__ = body.__doc__
        routed.__name__ = method
        setattr(cls, method, routed)
        return cls

    return deco


def same_shape(ctor, x, aux, params):
    return x


def module_op(kind: str, cls, schema: str, hip_impl: Callable, meta_impl: Callable, rebuild: Optional[Callable] = None):
    """A module-backed operator with a signature of its own (the streaming step: states cross the boundary too).
    `hip_impl(module, *args)` gets the live (or rebuilt) module; `params` and `cfg` must be the LAST two arguments."""
    def impl(*args):
        *rest, params, cfg = args
        return hip_impl(_resolve(kind, cfg, params), *rest)
    _KINDS[kind] = dict(cls=cls, method=None, shape=None, rebuild=rebuild or (lambda a: a))
    _define(kind, schema, impl, meta_impl)
    return getattr(getattr(torch.ops, NAMESPACE), kind)


def call_args(module: torch.nn.Module, kind: str):
    """(params, cfg) of a module-backed call, and the live-registry entry that lets the implementation find `module`."""
    params = module_tensors(module)
    cfg = json.dumps({"ctor": getattr(module, "_ctor_args", None)}, sort_keys=True)
    if params and params[0].device.type == "cuda":
        _LIVE[(kind, cfg, tuple(p.data_ptr() for p in params))] = module
    return params, cfg
