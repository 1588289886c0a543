"""out_conv shape only, one arithmetic / option set per process (for rocprofv3 --pmc passes): argv[1] in
{bf16x3, fp16x2, fp16x2_amax, fp16x2_bound0}"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puresound_amd import hip, _abi
dev = torch.device("cuda:0"); lib = _abi.lib()
N, T = 32, 3999; ldt = _abi.padded_frames(T)
K, M = 256, 512
which = sys.argv[1]
torch.manual_seed(0)
x = torch.randn(N, K, ldt, device=dev); w = torch.randn(M, K, device=dev) * 0.05
y = torch.empty(N, M, ldt, device=dev); r = torch.randn(N, M, ldt, device=dev); bias = torch.randn(M, device=dev)
g, b, sl = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.tensor([0.25], device=dev)
parts = lib.ps_dwconv_stats_parts(K, T)
st = torch.zeros(N, parts, 2, dtype=torch.float64, device=dev); st[:, 0, 1] = float(K * T)
p = hip.make_prologue(_abi.PS_NORM_GLOBAL, True, st, K * T, 1e-8, g, b, sl)
if which == "bf16x3":
    wb = hip.pack_wt_bf16(w, 3)
    run = lambda: hip.conv1x1_bf16(x, T, wb, M, p, bias, None, r, out=y)
else:
    wf, we = hip.pack_wt_f16x2(w)
    kw = dict(x_bound=1000.0)
    if which == "fp16x2_bound0":
        kw = {}
    run = lambda: hip.conv1x1_f16x2(x, T, wf, we, M, p, bias, None, r, out=y, want_amax=which == "fp16x2_amax", **kw)
for _ in range(6):
    run()
torch.cuda.synchronize()
