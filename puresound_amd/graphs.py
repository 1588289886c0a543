"""hipGraph replay of a whole `inference` call for small batches.

At one or a few utterances per call a Conv-TasNet forward is ~100 launches of 15-35 us each and a fifth of its wall
time is launch gaps.  Every kernel of the path is enqueued on the caller's stream through the C ABI and allocates
through torch's caching allocator only, so `torch.cuda.graph` (a hipGraph on ROCm) can capture the call as it is;
a replay then costs one launch.  (The streaming harness captures its frame step the same way,
puresound_amd/streaming/skim_inference.py.)

    fast = GraphedInference(model)           # SoTaskWrapModule / SiMoTaskWrapModule, on the GPU, eval()
    enhanced = fast(noisy)                    # first call per input shape: 3 eager warm-ups + capture; then replay

Results are bit-identical to the eager call.  One graph is kept per (input shapes, dtype); the graphs are dropped when
a parameter of the model is updated in place or `reset()` is called.
"""
from typing import Optional

import torch

from . import hip


def _cached_tensors(model: torch.nn.Module) -> list:
    """Every tensor the model's modules cache outside their parameters / buffers: scratch workspaces, packed-weight
    plans, identity tables.  A captured graph holds raw pointers into them, so the graph entry keeps them alive: a
    module may replace a cache entry later (a larger workspace for a larger batch, a new plan) without pulling memory
    from under an older graph."""
    seen, out = set(), []

    def walk(obj, depth):
        if torch.is_tensor(obj):
            if obj.is_cuda and id(obj) not in seen:
                seen.add(id(obj))
                out.append(obj)
        elif depth < 6:
            if isinstance(obj, dict):
                for v in obj.values():
                    walk(v, depth + 1)
            elif isinstance(obj, (list, tuple)):
                for v in obj:
                    walk(v, depth + 1)

    for m in model.modules():
        for k, v in m.__dict__.items():
            if k not in ("_parameters", "_buffers", "_modules"):
                walk(v, 0)
    walk(hip._EYE, 0)
    return out


class GraphedInference:
    def __init__(self, model: torch.nn.Module, max_graphs: int = 8) -> None:
        self.model = model
        self.max_graphs = max_graphs
        self._graphs = {}
        self._sig = None
        self._tensors = None
        self._switches = None

    def reset(self) -> None:
        """Drop the graphs (and re-read the model's parameter list: call it after REPLACING parameters; in-place updates
        such as load_state_dict or optimizer steps are noticed by themselves)."""
        self._graphs.clear()
        self._tensors = None
        self._stoch = None

    def _stochastic(self) -> bool:
        if getattr(self, "_stoch", None) is None:
            from .nnet.lobe.trivial import SpecAugment
            self._stoch = any(isinstance(m, SpecAugment) and (m.freq_mask > 0 or m.time_mask > 0)
                              for m in self.model.modules())
        return self._stoch

    def _signature(self):
        # per call, so it has to be cheap: the version counters of the tensors found at the first call (a walk over the
        # module tree costs more than the replay saves) and the arithmetic switches of the masker's blocks
        if self._tensors is None:
            self._tensors = list(self.model.parameters()) + list(self.model.buffers())
            masker = getattr(self.model, "masker", None)
            self._switches = [m for m in masker.modules() if hasattr(m, "gemm_precision")] if masker is not None else []
        return (tuple(t._version for t in self._tensors), tuple(m.gemm_precision for m in self._switches),
                getattr(self.model, "hip_streams", None), self.model.training)

    @torch.no_grad()
    def __call__(self, noisy: torch.Tensor, enroll: Optional[torch.Tensor] = None) -> torch.Tensor:
        hip.require_device(noisy, "GraphedInference")
        dev = noisy.device
        if self._stochastic():
            # SpecAugment draws its mask positions on the host at every call (the reference's layer masks in eval mode too,
            # lobe/trivial.py:7-58): a captured graph would replay ONE draw for ever.  Such a model runs eagerly.
            return self.model.inference(noisy) if enroll is None else self.model.inference(noisy, enroll)
        sig = self._signature()
        if sig != self._sig:
            self._graphs.clear()
            self._sig = sig
        key = (tuple(noisy.shape), noisy.dtype, None if enroll is None else tuple(enroll.shape))
        entry = self._graphs.get(key)
        if entry is None:
            if len(self._graphs) >= self.max_graphs:
                self._graphs.pop(next(iter(self._graphs)))
            static_in = noisy.clone()
            static_enroll = None if enroll is None else enroll.clone()
            args = (static_in,) if static_enroll is None else (static_in, static_enroll)
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):  # plans, scratch buffers and library handles exist before the capture
                for _ in range(3):
                    self.model.inference(*args)
            torch.cuda.current_stream(dev).wait_stream(side)
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                static_out = self.model.inference(*args)
            entry = (graph, static_in, static_enroll, static_out, _cached_tensors(self.model))
            self._graphs[key] = entry
        graph, static_in, static_enroll, static_out, _pinned = entry
        static_in.copy_(noisy)
        if static_enroll is not None:
            static_enroll.copy_(enroll)
        graph.replay()
        return static_out.clone()  # the reference contract: every call returns a fresh tensor
