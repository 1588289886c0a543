"""debug: whole-segment LSTM kernels (fp32 / fp16x2 recurrent product) against a float64 recurrence, by weight scale"""
import sys, torch, numpy as np
sys.path.insert(0, ".")
from puresound_amd import hip as H, _abi
torch.manual_seed(0)
dev = "cuda"
hid, n, k, s = 64, 2, 20, 9
for wscale in (1.0, 1e-1, 1e-2, 1e-3, 0.0):
    whh = (torch.rand(1, hid, 4 * hid) * 0.8 - 0.4) * wscale          # [D][H][4H] transposed
    gx = torch.randn(n, 4 * hid, s * k) * 1.5
    # float64 reference
    W = whh[0].double()
    g = gx.double().reshape(n, 4 * hid, s, k)
    h0 = torch.rand(n, hid, s) - 0.5; c0 = torch.rand(n, hid, s) - 0.5
    h = h0.double().permute(0, 2, 1).clone(); c = c0.double().permute(0, 2, 1).clone()
    ref = torch.zeros(n, hid, s, k, dtype=torch.float64)
    for t in range(k):
        a = g[..., t].permute(0, 2, 1) + h @ W
        i, f, gg, o = a[..., :hid], a[..., hid:2*hid], a[..., 2*hid:3*hid], a[..., 3*hid:]
        c = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h = torch.sigmoid(o) * torch.tanh(c)
        ref[..., t] = h.permute(0, 2, 1)
    ref = ref.reshape(n, hid, s * k)
    gxp = H.pad_rows(gx.to(dev))
    out = {}
    for name, flags, f in (("seg32", 4, False), ("grp32", 4 | 1 << 20, False), ("seg16x2", 4, True)):
        old = _abi.lib().ps_debug_flags(flags)
        ho, _ = H.lstm(gxp, whh.to(dev).contiguous(), hid, 1, s, k, k, 1, H.pad_rows(h0.to(dev)), H.pad_rows(c0.to(dev)), f16x2=f)
        torch.cuda.synchronize()
        _abi.lib().ps_debug_flags(old)
        out[name] = ho[..., :s * k].double().cpu()
    print(wscale, {kk: float((v - ref).abs().max()) for kk, v in out.items()})
