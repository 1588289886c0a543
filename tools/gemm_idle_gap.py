"""Is the fp16x2 GEMM's launch time a property of the kernel or of the power state it runs in?  The in_conv shape of
the benchmark, timed per launch (events) with idle gaps of 0 / 0.2 / 1 / 5 ms between launches (GPU box).
  PS_FLAGS=0x400000 python tools/gemm_idle_gap.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from puresound_amd import hip, _abi
dev = torch.device("cuda:0"); lib = _abi.lib()
flags = int(os.environ.get("PS_FLAGS", "0"), 0)
N, T = 32, 3999; ldt = _abi.padded_frames(T)
torch.manual_seed(0)
for name, K, M, pro, res in (("in ", 512, 256, False, False), ("out", 256, 512, True, True)):
    x = torch.randn(N, K, ldt, device=dev); w = torch.randn(M, K, device=dev) * 0.05
    if os.environ.get("ZEROS"):  # (same instructions, same bytes, no operand toggling)
        x.zero_(); w.zero_(); w[0, 0] = 1.0
    wb, we = hip.pack_wt_f16x2(w); y = torch.empty(N, M, ldt, device=dev)
    r = torch.randn(N, M, ldt, device=dev) if res else None
    bias = torch.randn(M, device=dev)
    g, b, sl = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.tensor([0.25], device=dev)
    st = torch.zeros(N, lib.ps_dwconv_stats_parts(K, T), 2, dtype=torch.float64, device=dev); st[:, 0, 1] = float(K * T)
    p = hip.make_prologue(_abi.PS_NORM_GLOBAL, True, st, K * T, 1e-8, g, b, sl) if pro else None
    kw = dict(x_bound=1000.0) if pro else dict(x_amax=hip.absmax(x, T))
    run = lambda: hip.conv1x1_f16x2(x, T, wb, we, M, p, bias, None, r, want_stats=not res, out=y, want_amax=res, **kw)
    lib.ps_debug_flags(flags)
    for gap_ms in (0.0, 1.0):
        for _ in range(5):
            run()
        torch.cuda.synchronize()
        ts = []
        for _ in range(40):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(); run(); e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3)
            if gap_ms:
                time.sleep(gap_ms * 1e-3)
        ts = np.array(ts[5:])
        print(f"flags {flags:#x} {name} gap {gap_ms:4.1f} ms: median {np.median(ts):6.1f} us  min {ts.min():6.1f}  max {ts.max():6.1f}", flush=True)
    lib.ps_debug_flags(0)
