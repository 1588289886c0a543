"""Per-workgroup timeline of ps_conv1x1_f32 (s_memtime stamps): phases and co-residency (GPU box)."""
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
    st = torch.zeros(N, 64, 2, dtype=torch.float64, device=dev); st[:, 0, 1] = float(K * T)
    p = hip.make_prologue(_abi.PS_NORM_GLOBAL, True, st, K * T, 1e-8, g, b, sl) if pro else None
    nwg = 32 * ((M + 255) // 256) * N
    buf = torch.zeros(nwg * 6, dtype=torch.int64, device=dev)
    for _ in range(3):
        hip.conv1x1(x, T, wt, M, p, bias, None, r, want_stats=not res, out=y)
    lib.ps_debug_buffer(buf.data_ptr())
    hip.conv1x1(x, T, wt, M, p, bias, None, r, want_stats=not res, out=y)
    torch.cuda.synchronize(); lib.ps_debug_buffer(None)
    s = buf.cpu().numpy().reshape(nwg, 6).astype(np.int64)
    t0 = s[:, 0].min()
    # s_memtime ticks at 100 MHz? report in ticks and as fractions
    dur = s[:, 3] - s[:, 0]; pro_t = s[:, 1] - s[:, 0]; loop_t = s[:, 2] - s[:, 1]; epi_t = s[:, 3] - s[:, 2]
    print(f"{name}: WGs={nwg} span={s[:,3].max()-t0} ticks; per-WG total med={np.median(dur):.0f} "
          f"prologue med={np.median(pro_t):.0f} loop med={np.median(loop_t):.0f} epilogue med={np.median(epi_t):.0f}")
    start = s[:, 0] - t0
    first = np.sort(start)[:512]; print("   start of first 512 WGs: max", first.max(), " later WGs start min", np.sort(start)[512:].min() if nwg > 512 else -1)
    # per-CU grouping: hw_id bits: cu_id [11:8], sh [12], se [15:13]; plus xcc
    cu = (s[:, 4] >> 8) & 0xff; key = s[:, 5] * 1000 + cu
    uniq, cnt = np.unique(key, return_counts=True)
    print("   distinct (xcc,cu-ish) keys:", len(uniq), " WGs per key min/max:", cnt.min(), cnt.max())
    # show timeline for one key
    k0 = uniq[0]; rows = s[key == k0]; rows = rows[np.argsort(rows[:, 0])]
    for rrow in rows[:8]:
        print("     ", [int(v - t0) for v in rrow[:4]])
