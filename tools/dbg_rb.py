"""Debug: the rb kernel (ps_debug_flags bit 22) against the exact-fp32 GEMM on one shape; where do they differ?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puresound_amd import hip, _abi
dev = torch.device("cuda:0"); lib = _abi.lib()
N, T = int(os.environ.get("N", 32)), 3999; ldt = _abi.padded_frames(T)
K, M, pro, res = 256, 512, True, True
torch.manual_seed(0)
x = torch.randn(N, K, ldt, device=dev); w = torch.randn(M, K, device=dev) * 0.05
wb, we = hip.pack_wt_f16x2(w)
r = torch.randn(N, M, ldt, device=dev); bias = torch.randn(M, device=dev)
g, b, sl = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.tensor([0.25], device=dev)
parts = lib.ps_dwconv_stats_parts(K, T)
st = torch.zeros(N, parts, 2, dtype=torch.float64, device=dev); st[:, 0, 1] = float(K * T)
p = hip.make_prologue(_abi.PS_NORM_GLOBAL, True, st, K * T, 1e-8, g, b, sl)
ref, _ = hip.conv1x1(x, T, hip.pack_wt(w), M, p, bias, None, r, want_stats=False)
for flags in (0, 1 << 22):
    lib.ps_debug_flags(flags)
    y = torch.full((N, M, ldt), float("nan"), device=dev)
    yy, _, amx = hip.conv1x1_f16x2(x, T, wb, we, M, p, bias, None, r, want_stats=False, out=y, want_amax=True, x_bound=1000.0)
    torch.cuda.synchronize(); lib.ps_debug_flags(0)
    d = (yy[:, :, :T] - ref[:, :, :T]).abs()
    bad = d > 1e-3
    print(f"flags {flags:#x}: max err {float(d.max()):.3e}  bad {int(bad.sum())} of {bad.numel()}  nan {int(torch.isnan(yy[:, :, :T]).sum())}")
    if bad.any():
        idx = bad.nonzero()
        print("  utterances", idx[:, 0].unique().tolist()[:40])
        print("  rows", idx[:, 1].unique().tolist()[:40], "...", int(idx[:, 1].unique().numel()))
        fr = idx[:, 2].unique()
        print("  frames", fr.tolist()[:40], "...", int(fr.numel()), "supertiles", (fr // 256).unique().tolist()[:40])
    a = amx.amax(1); t = yy[:, :, :T].abs().amax((1, 2))
    print("  amax err", float((a - t).abs().max()))
