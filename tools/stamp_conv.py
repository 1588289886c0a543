"""Per-workgroup timeline of ps_conv1x1_f32 (s_memtime stamps) -- persistent-kernel layout (GPU box)."""
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
    nwg = 512
    buf = torch.zeros(nwg * 6, dtype=torch.int64, device=dev)
    for _ in range(3):
        hip.conv1x1(x, T, wt, M, p, bias, None, r, want_stats=not res, out=y)
    lib.ps_debug_buffer(buf.data_ptr())
    hip.conv1x1(x, T, wt, M, p, bias, None, r, want_stats=not res, out=y)
    torch.cuda.synchronize(); lib.ps_debug_buffer(None)
    s = buf.cpu().numpy().reshape(nwg, 6).astype(np.int64)
    s = s[s[:, 0] > 0]
    t0 = s[:, 0].min()
    tiles = s[:, 3]; dur = s[:, 2] - s[:, 0]; fill = s[:, 1] - s[:, 0]
    nsteps = max(2, (K + 15) // 16)
    ideal = tiles * nsteps * 32 * 64 * 2  # two co-resident waves per SIMD share the MFMA pipe
    print(f"{name}: WGs={len(s)} span={s[:,2].max()-t0} cyc; tiles/WG {tiles.min()}..{tiles.max()}; per-WG total med={np.median(dur):.0f} "
          f"fill med={np.median(fill):.0f}; per tile med={np.median(dur/tiles):.0f} (MFMA-only 2-wave ideal {nsteps*32*64*2}); "
          f"start spread={np.ptp(s[:,0])}")
