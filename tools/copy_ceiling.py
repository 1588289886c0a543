"""What a plain device copy of one hidden map ([32, 256, 4224] fp32, 138 MB read + 138 MB written) achieves on this box,
next to ps_dwconv_f32 on the same buffers: the practical HBM ceiling for a 1 : 1 read / write stream."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puresound_amd import hip, _abi
dev = torch.device("cuda:0"); lib = _abi.lib()
N, H, T = 32, 256, 3999; ldt = _abi.padded_frames(T)
x = torch.randn(N, H, ldt, device=dev); y = torch.empty_like(x)
def timed(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3
us = timed(lambda: y.copy_(x))
print(f"torch copy_ [32,256,4224] fp32: {us:.1f} us = {2 * x.numel() * 4 / us / 1e6:.2f} TB/s (read + write)")
us = timed(lambda: torch.add(x, 1.0, out=y))
print(f"torch add(x, 1) out=y: {us:.1f} us = {2 * x.numel() * 4 / us / 1e6:.2f} TB/s")
big = torch.randn(N, 512, ldt, device=dev); yb = torch.empty_like(big)
us = timed(lambda: yb.copy_(big))
print(f"torch copy_ [32,512,4224] fp32: {us:.1f} us = {2 * big.numel() * 4 / us / 1e6:.2f} TB/s")
w = torch.randn(H, 1, 3, device=dev) * 0.3; b = torch.randn(H, device=dev)
g, be, sl = torch.rand(H, device=dev) + 0.5, torch.randn(H, device=dev) * 0.1, torch.tensor([0.25], device=dev)
parts = lib.ps_conv1x1_stats_parts(H, T)
st = torch.zeros(N, parts, 2, dtype=torch.float64, device=dev); st[:, 0, 1] = float(H * T)
p = hip.make_prologue(_abi.PS_NORM_GLOBAL, True, st, H * T, 1e-8, g, be, sl)
for d in (1, 16, 128):
    us = timed(lambda: hip.dwconv(x, T, w, b, d, d, p, want_stats=True))
    print(f"ps_dwconv_f32 dilation {d}: {us:.1f} us = {2 * N * H * T * 4 / us / 1e6:.2f} TB/s algorithmic")
