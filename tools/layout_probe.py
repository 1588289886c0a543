"""Does DRAM locality matter to the GEMMs?  The same bytes and FLOPs as the 32-utterance launches, once in the row layout
of the path ([32][C][4224]: a K-step reads sixteen 512-byte pieces 16.9 KB apart) and once as 512 "utterances" of 256
frames ([512][C][256]: the sixteen rows of a K-step, both halves' tiles, are one contiguous 16 KiB) -- what a tile-major
activation layout would give the kernel.  Run on the GPU box."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puresound_amd import hip, _abi
dev = torch.device("cuda:0"); lib = _abi.lib()
torch.manual_seed(0)
for name, K, M, pro, res in (("in ", 512, 256, False, False), ("pw ", 256, 256, True, False), ("out", 256, 512, True, True)):
    line = name
    for N, T, ldt in ((32, 4096, 4224), (512, 256, 256)):
        x = torch.randn(N, K, ldt, device=dev); w = torch.randn(M, K, device=dev) * 0.05
        y = torch.empty(N, M, ldt, device=dev); r = torch.randn(N, M, ldt, device=dev) if res else None
        bias = torch.randn(M, device=dev)
        g, b, sl = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.tensor([0.25], device=dev)
        st = torch.zeros(N, lib.ps_dwconv_stats_parts(K, T), 2, dtype=torch.float64, device=dev); st[:, 0, 1] = float(K * T)
        p = hip.make_prologue(_abi.PS_NORM_GLOBAL, True, st, K * T, 1e-8, g, b, sl) if pro else None
        wf, we = hip.pack_wt_f16x2(w)
        kw = dict(x_bound=1000.0) if pro else dict(x_amax=hip.absmax(x, T))
        run = lambda: hip.conv1x1_f16x2(x, T, wf, we, M, p, bias, None, r, want_stats=not res, out=y, want_amax=res, **kw)
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        line += f"   [{N}][{K}->{M}][{T} of {ldt}]: {e0.elapsed_time(e1) / 20 * 1e3:.0f} us"
    print(line, flush=True)
