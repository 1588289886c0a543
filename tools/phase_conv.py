"""Where a conv1x1 workgroup's cycles go (needs a library built with -DPS_PHASE_STAMPS; GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from puresound_amd import hip, _abi
dev = torch.device("cuda:0"); lib = _abi.lib()
N, T = 32, 3999; ldt = _abi.padded_frames(T)
shapes = {"in": (512, 256, False, False), "pw": (256, 256, True, False), "out": (256, 512, True, True)}
for name, (K, M, pro, res) in shapes.items():
    x = torch.randn(N, K, ldt, device=dev); wt = hip.pack_wt(torch.randn(M, K, device=dev) * 0.05)
    y = torch.empty(N, M, ldt, device=dev); r = torch.randn(N, M, ldt, device=dev) if res else None
    bias = torch.randn(M, device=dev)
    g, b, sl = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.tensor([0.25], device=dev)
    parts = lib.ps_dwconv_stats_parts(K, T)
    st = torch.zeros(N, parts, 2, dtype=torch.float64, device=dev); st[:, 0, 1] = float(K * T)
    p = hip.make_prologue(_abi.PS_NORM_GLOBAL, True, st, K * T, 1e-8, g, b, sl) if pro else None
    buf = torch.zeros(512 * 6, dtype=torch.int64, device=dev)
    for _ in range(3):
        hip.conv1x1(x, T, wt, M, p, bias, None, r, want_stats=not res, out=y)
    lib.ps_debug_buffer(buf.data_ptr())
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    hip.conv1x1(x, T, wt, M, p, bias, None, r, want_stats=not res, out=y)
    e1.record()
    torch.cuda.synchronize(); lib.ps_debug_buffer(None)
    us = e0.elapsed_time(e1) * 1e3
    s = buf.cpu().numpy().reshape(512, 6).astype(np.float64)
    s = s[s[:, 0] > 0]
    tiles = s[:, 3]; nsteps = max(2, (K + 15) // 16); steps = tiles * nsteps
    f = lambda c: np.median(s[:, c] / steps)
    print(f"{name}: kernel {us:.0f} us; WG cycles med {np.median(s[:,0]):.0f} max {s[:,0].max():.0f} -> implied clock {s[:,0].max()/us/1e3:.2f} GHz; tiles/WG {tiles.min():.0f}..{tiles.max():.0f}")
    kmaj = np.median(s[:, 2] / (steps - tiles)); last = np.median(s[:, 5] / tiles)
    print(f"{name}: per K-step (wave 0, median): total {np.median(s[:,0]/steps):.0f} = vmcnt-wait {f(1):.0f} + barrier+dma-issue {f(4):.0f} "
          f"+ body; k-major body {kmaj:.0f}/step, LAST-step body {last:.0f}/tile   (64 MFMAs = 4096 alone, 8192 shared)")
