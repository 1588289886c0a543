"""The three Conv-TasNet GEMM shapes at 32 utterances in the fp16x2 arithmetic, a few launches each (for rocprofv3 --pmc)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puresound_amd import hip, _abi
dev = torch.device("cuda:0"); lib = _abi.lib()
N, T = 32, 3999; ldt = _abi.padded_frames(T)
torch.manual_seed(0)
for K, M, pro, res in ((512, 256, False, False), (256, 256, True, False), (256, 512, True, True)):
    x = torch.randn(N, K, ldt, device=dev); w = torch.randn(M, K, device=dev) * 0.05
    y = torch.empty(N, M, ldt, device=dev); r = torch.randn(N, M, ldt, device=dev) if res else None
    bias = torch.randn(M, device=dev)
    g, b, sl = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.tensor([0.25], device=dev)
    st = torch.zeros(N, lib.ps_dwconv_stats_parts(K, T), 2, dtype=torch.float64, device=dev); st[:, 0, 1] = float(K * T)
    p = hip.make_prologue(_abi.PS_NORM_GLOBAL, True, st, K * T, 1e-8, g, b, sl) if pro else None
    wf, we = hip.pack_wt_f16x2(w)
    kw = dict(x_bound=1000.0) if pro else dict(x_amax=hip.absmax(x, T))
    for _ in range(5):
        hip.conv1x1_f16x2(x, T, wf, we, M, p, bias, None, r, want_stats=not res, out=y, want_amax=res, **kw)
    torch.cuda.synchronize()
