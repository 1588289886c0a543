"""Timing of the HBM-bound kernels of config 2 at 32 utterances (run on the GPU box): dwconv per dilation, the learned
encoder and decoder.  Prints microseconds per launch and algorithmic TB/s."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puresound_amd import hip, _abi

dev = torch.device("cuda:0"); lib = _abi.lib()
N, T, H, C = 32, 3999, 256, 512
ldt = _abi.padded_frames(T)
torch.manual_seed(0)


import ctypes


def timeit(fn, family, reps=20):
    """microseconds per launch of kernel family `family`, from the library's own hipEvents around the launch (the
    torch-side wrapper also allocates and zero-fills its outputs: not kernel time)."""
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    lib.ps_profile_enable(1)
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    lib.ps_profile_enable(0)
    ms, cnt = ctypes.c_double(), ctypes.c_int()
    _abi.check(lib.ps_profile_read(family.encode(), ctypes.byref(ms), ctypes.byref(cnt)), "ps_profile_read")
    return ms.value / max(cnt.value, 1) * 1e3


x = torch.randn(N, H, ldt, device=dev)
w, b = torch.randn(H, 1, 3, device=dev), torch.randn(H, device=dev)
g, be, sl = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.1, torch.tensor([0.25], device=dev)
parts = lib.ps_conv1x1_stats_parts(H, T)
st = torch.zeros(N, parts, 2, dtype=torch.float64, device=dev); st[:, 0, 1] = float(H * T)
pro = hip.make_prologue(_abi.PS_NORM_GLOBAL, True, st, H * T, 1e-8, g, be, sl)
byts = 2.0 * N * H * T * 4
res = []
for d in (1, 2, 4, 8, 16, 32, 64, 128):
    us = timeit(lambda: hip.dwconv(x, T, w, b, d, d, pro, True), "dwconv")
    res.append(f"d{d}={us:.1f}us({byts / us / 1e6:.2f}TB/s)")
print("dwconv", " ".join(res), flush=True)
for name, p_, st_ in (("no-norm,stats", None, True), ("norm,no-stats", pro, False), ("no-norm,no-stats", None, False)):
    us = timeit(lambda: hip.dwconv(x, T, w, b, 8, 8, p_, st_), "dwconv")
    print(f"dwconv d8 [{name}] {us:.1f}us ({byts / us / 1e6:.2f} TB/s)", flush=True)
wav = (torch.rand(N, 64000, device=dev) - 0.5)
we = torch.randn(C, 1, 32, device=dev) * 0.1
us = timeit(lambda: hip.free_encode(wav, we, 16, False), "free_encode")
print(f"free_encode {us:.1f}us ({N * (C * T + 64000) * 4 / us / 1e6:.2f} TB/s)", flush=True)
feats, mask = torch.randn(N, C, ldt, device=dev), torch.randn(N, C, ldt, device=dev)
us = timeit(lambda: hip.free_decode(feats, T, we, 16, mask, "relu", "linear"), "free_decode")
print(f"free_decode {us:.1f}us ({N * (2 * C * T + 64000) * 4 / us / 1e6:.2f} TB/s)", flush=True)
