"""Cycle buckets of conv1x1_bf16_il_kernel per (workgroup, half) from a -DPS_PP_STAMPS build (s_memtime):
interval body (MFMAs + chunks) / wait + barrier / drain.  Run on the GPU box:
  make -C puresound_amd/csrc clean all EXTRA=-DPS_PP_STAMPS && python tools/stamp_il.py [extra debug flags]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from puresound_amd import hip, _abi
dev = torch.device("cuda:0"); lib = _abi.lib()
N, T = 32, 3999; ldt = _abi.padded_frames(T)
shapes = {"in": (512, 256, False, False), "pw": (256, 256, True, False), "out": (256, 512, True, True)}
extra = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
lib.ps_debug_flags(extra)
for planes in (3,):
    for name, (K, M, pro, res) in shapes.items():
        x = torch.randn(N, K, ldt, device=dev); wb = hip.pack_wt_bf16(torch.randn(M, K, device=dev) * 0.05, planes)
        y = torch.empty(N, M, ldt, device=dev); r = torch.randn(N, M, ldt, device=dev) if res else None
        bias = torch.randn(M, device=dev)
        g, b, sl = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.tensor([0.25], device=dev)
        parts = lib.ps_dwconv_stats_parts(K, T)
        st = torch.zeros(N, parts, 2, dtype=torch.float64, device=dev); st[:, 0, 1] = float(K * T)
        p = hip.make_prologue(_abi.PS_NORM_GLOBAL, True, st, K * T, 1e-8, g, b, sl) if pro else None
        buf = torch.zeros(512 * 6, dtype=torch.int64, device=dev)
        for _ in range(3):
            hip.conv1x1_bf16(x, T, wb, M, p, bias, None, r, want_stats=not res, out=y)
        lib.ps_debug_buffer(buf.data_ptr())
        hip.conv1x1_bf16(x, T, wb, M, p, bias, None, r, want_stats=not res, out=y)
        torch.cuda.synchronize(); lib.ps_debug_buffer(None)
        s = buf.cpu().numpy().reshape(512, 6).astype(np.int64)
        for h in (0, 1):
            q = s[h::2]
            tot = np.maximum(q[:, 5], 1)
            tiles = tot / ((K + 15) // 16)
            print(f"bf16x{planes} {name} [half {h}]: steps/WG {int(np.median(tot))} total cyc {int(np.median(q[:,0]))} = {np.median(q[:,0]/tot):.0f}/step; "
                  f"per step med: body {np.median(q[:,1]/tot):.0f} stage {np.median(q[:,2]/tot):.0f} wait+barrier {np.median(q[:,3]/tot):.0f}; "
                  f"drain per tile {np.median(q[:,4]/tiles):.0f}", flush=True)
