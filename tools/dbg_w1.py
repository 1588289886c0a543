"""w1 kernel debug: small shapes through ps_conv1x1_f16x2_f32 (bit 28 = persistent kernel at any size), error map per 32 x 32 block."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puresound_amd import hip, _abi
dev = torch.device("cuda:0"); lib = _abi.lib()
torch.manual_seed(0)
def run(N, K, M, T, flags, pro=False, res=False, stats=True):
    ldt = _abi.padded_frames(T)
    x = torch.randn(N, K, ldt, device=dev); w = torch.randn(M, K, device=dev) * 0.05
    bias = torch.randn(M, device=dev)
    r = torch.randn(N, M, ldt, device=dev) if res else None
    wb, we = hip.pack_wt_f16x2(w)
    p = None
    if pro:
        g, b, sl = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.tensor([0.25], device=dev)
        st = torch.zeros(N, lib.ps_dwconv_stats_parts(K, T), 2, dtype=torch.float64, device=dev); st[:, 0, 1] = float(K * T)
        p = hip.make_prologue(_abi.PS_NORM_GLOBAL, True, st, K * T, 1e-8, g, b, sl)
        xin = torch.nn.functional.prelu(x * g[None, :, None] + b[None, :, None], sl)
    else:
        xin = x
    kw = dict(x_bound=1000.0) if pro else dict(x_amax=hip.absmax(x, T))
    lib.ps_debug_flags(flags)
    y, st2, am = hip.conv1x1_f16x2(x, T, wb, we, M, p, bias, None, r.clone() if res else None, want_stats=stats and not res, want_amax=True, **kw)
    torch.cuda.synchronize(); lib.ps_debug_flags(0)
    ref = torch.einsum("mk,nkt->nmt", w.double(), xin.double()) + bias.double()[None, :, None]
    if res: ref = ref + r.double()
    d = (y[:, :, :T].double() - ref[:, :, :T]).abs()
    print(f"N={N} K={K} M={M} T={T} flags={flags:#x} pro={pro} res={res}: max err {float(d.max()):.3e} (ref max {float(ref.abs().max()):.2f}) nan={int(torch.isnan(y[:, :, :T]).sum())}")
    if float(d.max()) > 1e-3 or torch.isnan(d).any():
        d = torch.nan_to_num(d, nan=9e9)
        tb = (T + 31) // 32
        pad = torch.zeros(N, M, tb * 32, device=dev, dtype=torch.float64); pad[:, :, :T] = d
        blk = pad.view(N, M // 32, 32, tb, 32).amax(dim=(2, 4))
        for n in range(N):
            print(f" utterance {n}: rows = 32-row blocks, cols = 32-frame blocks (x = bad)")
            for i in range(M // 32):
                print("  " + "".join("x" if v > 1e-3 else "." for v in blk[n, i].tolist()))
    if st2 is not None:
        s_ref = torch.stack([ref[:, :, :T].sum((1, 2)), (ref[:, :, :T] ** 2).sum((1, 2))], 1)
        print("   stats rel err", float(((st2.sum(1) - s_ref).abs() / s_ref.abs()).max()))
    if am is not None:
        print("   amax err", float((am.max(1).values.double() - ref[:, :, :T].abs().amax((1, 2))).abs().max()))
B28 = 1 << 28
W1 = B28 | 128
for args in [(2, 64, 256, 300, W1), (2, 64, 256, 300, B28), (2, 256, 512, 700, W1), (3, 128, 256, 515, W1)]:
    run(*args)
run(2, 128, 256, 515, W1, pro=True)
run(2, 128, 512, 515, W1, pro=True, res=True)
