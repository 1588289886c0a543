"""fp16x2 GEMM on the three Conv-TasNet shapes at the benchmark's batch: time per launch, algorithmic TB/s (SURVEY 8d
bytes), a check against the exact-fp32 kernel, and -- on a -DPS_PP_STAMPS build -- the s_memtime buckets.
  PURESOUND_HIP_LIB=tools/_variants/NAME.so python tools/time_f16x2.py [debug flags] [--stamps] [--nocheck]
debug flags (PS_PP_STAMPS builds): bit 24 no MFMA, bit 25 DMA without traffic, bit 26 no side-work chunks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from puresound_amd import hip, _abi

dev = torch.device("cuda:0"); lib = _abi.lib()
args = [a for a in sys.argv[1:] if not a.startswith("--")]
flags = int(args[0], 0) if args else 0
stamps = "--stamps" in sys.argv
N, T = 32, 3999; ldt = _abi.padded_frames(T)
shapes = {"in": (512, 256, False, False), "pw": (256, 256, True, False), "out": (256, 512, True, True)}
tag = os.path.basename(os.environ.get("PURESOUND_HIP_LIB", "default"))
torch.manual_seed(0)
line = []
for name, (K, M, pro, res) in shapes.items():
    x = torch.randn(N, K, ldt, device=dev)
    w = torch.randn(M, K, device=dev) * 0.05
    if "--zeros" in sys.argv:  # (how much of the time is the energy of toggling operands)
        x.zero_(); w.zero_(); w[0, 0] = 1.0
    wb, we = hip.pack_wt_f16x2(w)
    y = torch.empty(N, M, ldt, device=dev)
    r = torch.randn(N, M, ldt, device=dev) if res else None
    bias = torch.randn(M, device=dev)
    g, b, sl = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.tensor([0.25], device=dev)
    parts = lib.ps_dwconv_stats_parts(K, T)
    st = torch.zeros(N, parts, 2, dtype=torch.float64, device=dev); st[:, 0, 1] = float(K * T)
    p = hip.make_prologue(_abi.PS_NORM_GLOBAL, True, st, K * T, 1e-8, g, b, sl) if pro else None
    kw = dict(x_bound=1000.0) if pro else dict(x_amax=hip.absmax(x, T))
    run = lambda: hip.conv1x1_f16x2(x, T, wb, we, M, p, bias, None, r, want_stats=not res, out=y, want_amax=res, **kw)
    err = float("nan")
    if "--nocheck" not in sys.argv and not (flags >> 24):
        lib.ps_debug_flags(flags & 0xffffff)  # (kernel-variant bits stay: the check runs the kernel that is timed)
        yy, st2, amx = run()
        lib.ps_debug_flags(0)
        ref, st_ref = hip.conv1x1(x, T, hip.pack_wt(w), M, p, bias, None, r, want_stats=not res)
        err = float((yy[:, :, :T] - ref[:, :, :T]).abs().max() / ref[:, :, :T].abs().max())
        if amx is not None:
            err = max(err, float((amx.amax(1) - yy[:, :, :T].abs().amax((1, 2))).abs().max() / yy[:, :, :T].abs().max()))
        if st2 is not None:
            err = max(err, float((st2.sum(1) - st_ref.sum(1)).abs().max() / st_ref.sum(1).abs().max()))
    lib.ps_debug_flags(flags)
    for _ in range(5):
        run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    reps = 30
    e0.record()
    for _ in range(reps):
        run()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / reps * 1e3
    alg = N * T * 4 * (K + M + (M if res else 0))
    line.append(f"{name} {us:6.1f} us {alg / us / 1e6:5.2f} TB/s err {err:.1e}")
    if stamps:
        buf = torch.zeros(512 * 6, dtype=torch.int64, device=dev)
        lib.ps_debug_buffer(buf.data_ptr())
        run()
        torch.cuda.synchronize(); lib.ps_debug_buffer(None)
        s = buf.cpu().numpy().reshape(512, 6).astype(np.int64)
        for h in (0, 1):
            q = s[h::2]
            tot = np.maximum(q[:, 5], 1)
            tiles = tot / ((K + 15) // 16)
            print(f"   {name} [half {h}] clock {np.median(q[:, 0] / np.maximum(q[:, 2], 1)) / 10:.2f} GHz total cyc {int(np.median(q[:, 0]))} = "
                  f"{np.median(q[:, 0] / tot):.0f}/step; [il: body | w1: vmcnt wait] {np.median(q[:, 1] / tot):.0f} [il: wait+barrier | w1: barrier] {np.median(q[:, 3] / tot):.0f}; "
                  f"drain per tile {np.median(q[:, 4] / tiles):.0f}", flush=True)
    lib.ps_debug_flags(0)
print(f"{tag} flags={flags:#x}: " + " | ".join(line), flush=True)
