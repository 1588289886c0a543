"""fp16x2 GEMM (ps_conv1x1_f16x2_f32): error against an fp64 product, next to the exact-fp32 and bf16x3 kernels, and launch
times on the three Conv-TasNet shapes (run on the GPU box)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from puresound_amd import hip, _abi

dev = torch.device("cuda:0")
lib = _abi.lib()
T = 3999
ldt = _abi.padded_frames(T)
shapes = {"in  K512 M256": (512, 256, False, False), "pw  K256 M256": (256, 256, True, False),
          "out K256 M512": (256, 512, True, True)}
torch.manual_seed(0)


def make(N, K, M, pro, res, xs=1.0):
    x = torch.randn(N, K, ldt, device=dev) * xs
    w = torch.randn(M, K, device=dev) * 0.05
    r = torch.randn(N, M, ldt, device=dev) if res else None
    bias = torch.randn(M, device=dev)
    g, b, sl = torch.rand(K, device=dev) + 0.5, torch.randn(K, device=dev) * 0.1, torch.tensor([0.25], device=dev)
    p = None
    if pro:
        parts = lib.ps_dwconv_stats_parts(K, T)
        st = torch.zeros(N, parts, 2, dtype=torch.float64, device=dev)
        xx = x[:, :, :T].double()
        st[:, 0, 0] = xx.sum((1, 2))
        st[:, 0, 1] = (xx * xx).sum((1, 2))
        p = hip.make_prologue(_abi.PS_NORM_GLOBAL, True, st, K * T, 1e-8, g, b, sl)
    return x, w, r, bias, (g, b, sl, st if pro else None), p


def ref64(x, w, r, bias, gbs, pro):
    g, b, sl, _ = gbs
    u = x[:, :, :T].double()
    if pro:
        mean = u.mean((1, 2), keepdim=True)
        var = (u * u).mean((1, 2), keepdim=True) - mean * mean
        u = (u - mean) / torch.sqrt(var + 1e-8) * g.double()[None, :, None] + b.double()[None, :, None]
        u = torch.where(u >= 0, u, u * sl.double())
    y = torch.einsum("mk,nkt->nmt", w.double(), u) + bias.double()[None, :, None]
    if r is not None:
        y = y + r[:, :, :T].double()
    return y


print("== error against an fp64 product (2 utterances): max |d| / rms(y), rms(d) / rms(y)")
for xs in (1.0, 1e-3, 300.0):
    for name, (K, M, pro, res) in shapes.items():
        x, w, r, bias, gbs, p = make(2, K, M, pro, res, xs)
        y64 = ref64(x, w, r, bias, gbs, pro)
        rms = float(y64.pow(2).mean().sqrt())
        outs = {}
        outs["fp32"] = hip.conv1x1(x, T, hip.pack_wt(w), M, p, bias, None, r, want_stats=not res)[0]
        outs["bf16x3"] = hip.conv1x1_bf16(x, T, hip.pack_wt_bf16(w, 3), M, p, bias, None, r, want_stats=not res)[0]
        wf, we = hip.pack_wt_f16x2(w)
        lib.ps_debug_flags(0)
        outs["fp16x2/default range"] = hip.conv1x1_f16x2(x, T, wf, we, M, p, bias, None, r, want_stats=not res)[0]
        # the range a caller would give: a bound behind the norm, the producer's maxima for raw rows
        kw = dict(x_bound=float(gbs[0].abs().max()) * (K * T) ** 0.5 + float(gbs[1].abs().max())) if pro else dict(x_amax=hip.absmax(x, T))
        outs["fp16x2"], _, am = hip.conv1x1_f16x2(x, T, wf, we, M, p, bias, None, r, want_stats=not res, want_amax=True, **kw)
        lib.ps_debug_flags(1 << 27)
        outs["fp16x2/simple"], _, am2 = hip.conv1x1_f16x2(x, T, wf, we, M, p, bias, None, r, want_stats=not res, want_amax=True, **kw)
        lib.ps_debug_flags(0)
        ymax = outs["fp16x2"][:, :, :T].abs().amax((1, 2))
        assert torch.equal(am.amax(1), ymax) and torch.equal(am2.amax(1), outs["fp16x2/simple"][:, :, :T].abs().amax((1, 2))), (am.amax(1), ymax)
        line = f"x scale {xs:g} {name}:"
        for k, y in outs.items():
            d = y[:, :, :T].double() - y64
            line += f"  {k} {float(d.abs().max()) / rms:.2e} / {float(d.pow(2).mean().sqrt()) / rms:.2e}"
        print(line, flush=True)

x = torch.randn(32, 512, ldt, device=dev)
for _ in range(3):
    hip.absmax(x, T)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10):
    hip.absmax(x, T)
e1.record()
torch.cuda.synchronize()
print(f"absmax of 32 x 512 x {T}: {e0.elapsed_time(e1) / 10 * 1e3:.0f} us")
print("== launch times, 32 utterances")
for name, (K, M, pro, res) in shapes.items():
    x, w, r, bias, gbs, p = make(32, K, M, pro, res)
    y = torch.empty(32, M, ldt, device=dev)
    wf, we = hip.pack_wt_f16x2(w)
    w3 = hip.pack_wt_bf16(w, 3)
    kw = dict(x_bound=1000.0) if pro else dict(x_amax=hip.absmax(x, T))
    line = name + ":"
    for label, fn in (("bf16x3", lambda: hip.conv1x1_bf16(x, T, w3, M, p, bias, None, r, want_stats=not res, out=y)),
                      ("fp16x2", lambda: hip.conv1x1_f16x2(x, T, wf, we, M, p, bias, None, r, want_stats=not res, out=y)),
                      ("fp16x2+amax", lambda: hip.conv1x1_f16x2(x, T, wf, we, M, p, bias, None, r, want_stats=not res, out=y, want_amax=True, **kw))):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        line += f"  {label} {e0.elapsed_time(e1) / 20 * 1e3:.0f} us"
    print(line, flush=True)
