"""Ablation timing of ps_conv1x1_f32 on the three Conv-TasNet GEMM shapes (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puresound_amd import hip, _abi

dev = torch.device("cuda:0")
lib = _abi.lib()
N, T = 32, 3999
ldt = _abi.padded_frames(T)
shapes = {"in  K512 M256": (512, 256, False, False), "pw  K256 M256": (256, 256, True, False),
          "out K256 M512": (256, 512, True, True)}
flags = {"full": 0}
torch.manual_seed(0)
for name, (K, M, pro, res) in shapes.items():
    x = torch.randn(N, K, ldt, device=dev)
    wt = hip.pack_wt(torch.randn(M, K, device=dev) * 0.05)
    y = torch.empty(N, M, ldt, device=dev)
    r = torch.randn(N, M, ldt, device=dev) if res else None
    bias = torch.randn(M, device=dev)
    g, b, sl = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.tensor([0.25], device=dev)
    parts = lib.ps_dwconv_stats_parts(K, T)
    st = torch.zeros(N, parts, 2, dtype=torch.float64, device=dev)
    st[:, 0, 0] = 0.0
    st[:, 0, 1] = float(K * T)
    p = hip.make_prologue(_abi.PS_NORM_GLOBAL, True, st, K * T, 1e-8, g, b, sl) if pro else None
    flop = 2.0 * N * T * K * M
    res_line = []
    for fname, f in flags.items():
        lib.ps_debug_flags(f)
        for _ in range(3):
            hip.conv1x1(x, T, wt, M, p, bias, None, r, want_stats=not res, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 10
        e0.record()
        for _ in range(reps):
            hip.conv1x1(x, T, wt, M, p, bias, None, r, want_stats=not res, out=y)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / reps * 1e3
        res_line.append(f"{fname}={us:.0f}us")
    lib.ps_debug_flags(0)
    # flags: 0 = default (interleaved one-barrier persistent kernel), 32 = two-barrier ping-pong kernel, bit 27 = simple kernel,
    # bit 30 = single-wave experiment; extra flag bits from the command line are OR-ed in (kernel experiments)
    extra = int(sys.argv[1], 0) if len(sys.argv) > 1 else 0
    for planes, abl in ((3, 0), (3, 32), (1, 0), (1, 32)):
        lib.ps_debug_flags(abl | (extra if abl == 0 else 0))
        wb = hip.pack_wt_bf16(torch.randn(M, K, device=dev) * 0.05, planes)
        for _ in range(3):
            hip.conv1x1_bf16(x, T, wb, M, p, bias, None, r, want_stats=not res, out=y)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            hip.conv1x1_bf16(x, T, wb, M, p, bias, None, r, want_stats=not res, out=y)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 10 * 1e3
        res_line.append(f"bf16x{planes}/{ {0: 'il', 32: 'pp'}.get(abl, 'simple') }={us:.0f}us ({flop / us / 1e6 * (6 if planes == 3 else 1) / 2500:.2f})")
    lib.ps_debug_flags(0)
    wf, we = hip.pack_wt_f16x2(torch.randn(M, K, device=dev) * 0.05)
    kw = dict(x_bound=1000.0) if pro else dict(x_amax=hip.absmax(x, T))
    for _ in range(3):
        hip.conv1x1_f16x2(x, T, wf, we, M, p, bias, None, r, want_stats=not res, out=y, want_amax=res, **kw)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        hip.conv1x1_f16x2(x, T, wf, we, M, p, bias, None, r, want_stats=not res, out=y, want_amax=res, **kw)
    e1.record()
    torch.cuda.synchronize()
    res_line.append(f"fp16x2/il={e0.elapsed_time(e1) / 20 * 1e3:.0f}us")
    print(name, f"(peak {flop / 157.3e12 * 1e6:.0f}us)", "  ".join(res_line), flush=True)
