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
for planes in (3, 2, 1):
    for name, (K, M, pro, res) in shapes.items():
        x = torch.randn(N, K, ldt, device=dev)
        if planes == 2:
            wb, we = hip.pack_wt_f16x2(torch.randn(M, K, device=dev) * 0.05)
            am = hip.absmax(x, T)
            kw = dict(x_bound=1000.0) if pro else dict(x_amax=am)
            run = lambda: hip.conv1x1_f16x2(x, T, wb, we, M, p, bias, None, r, want_stats=not res, out=y, want_amax=res, **kw)
        else:
            wb = hip.pack_wt_bf16(torch.randn(M, K, device=dev) * 0.05, planes)
            run = lambda: hip.conv1x1_bf16(x, T, wb, M, p, bias, None, r, want_stats=not res, out=y)
        y = torch.empty(N, M, ldt, device=dev); r = torch.randn(N, M, ldt, device=dev) if res else None
        bias = torch.randn(M, device=dev)
        g, b, sl = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.tensor([0.25], device=dev)
        parts = lib.ps_dwconv_stats_parts(K, T)
        st = torch.zeros(N, parts, 2, dtype=torch.float64, device=dev); st[:, 0, 1] = float(K * T)
        p = hip.make_prologue(_abi.PS_NORM_GLOBAL, True, st, K * T, 1e-8, g, b, sl) if pro else None
        buf = torch.zeros(512 * 6, dtype=torch.int64, device=dev)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        lib.ps_debug_buffer(buf.data_ptr())
        run()
        torch.cuda.synchronize(); lib.ps_debug_buffer(None)
        s = buf.cpu().numpy().reshape(512, 6).astype(np.int64)
        for h in (0, 1):
            q = s[h::2]
            tot = np.maximum(q[:, 5], 1)
            tiles = tot / ((K + 15) // 16)
            print(f"{'fp16' if planes == 2 else 'bf16'}x{planes} {name} {us:.0f} us, clock {np.median(q[:,0]) / us / 1e3:.2f} GHz [half {h}]: steps/WG {int(np.median(tot))} total cyc {int(np.median(q[:,0]))} = {np.median(q[:,0]/tot):.0f}/step; "
                  f"per step med: body {np.median(q[:,1]/tot):.0f} stage {np.median(q[:,2]/tot):.0f} wait+barrier {np.median(q[:,3]/tot):.0f}; "
                  f"drain per tile {np.median(q[:,4]/tiles):.0f}", flush=True)
