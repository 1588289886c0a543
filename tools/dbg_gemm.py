import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from puresound_amd import hip as H, _abi
dev = torch.device("cuda:0")
flags = int(sys.argv[1], 0) if len(sys.argv) > 1 else (1 << 28)
n, k, m, t = 2, 256, 256, 500
torch.manual_seed(0)
x = torch.randn(n, k, t); w = torch.randn(m, k) * 0.1; b = torch.randn(m)
ref = torch.matmul(w.double(), x.double()) + b.double().reshape(1, -1, 1)
_abi.lib().ps_debug_flags(flags)
y, st = H.conv1x1_bf16(H.pad_rows(x.to(dev)), t, H.pack_wt_bf16(w.to(dev), 3), m, None, b.to(dev), None, None, want_stats=True)
torch.cuda.synchronize()
_abi.lib().ps_debug_flags(0)
err = (y[..., :t].cpu().double() - ref).abs().numpy()
print("max err", err.max(), "ref max", ref.abs().max().item())
bad = err > 1e-3
print("bad fraction", bad.mean())
for nn in range(n):
    e = bad[nn]
    print("utt", nn, "bad rows by 8:", [int(e[r:r + 8].any()) for r in range(0, m, 8)])
    print("utt", nn, "bad cols by 32:", [int(e[:, c:c + 32].any()) for c in range(0, t, 32)])
r, c = np.argwhere(bad[0])[0] if bad[0].any() else (0, 0)
print("first bad", r, c, y[0, r, c].item(), ref[0, r, c].item())
yy = y[0, :, :t].cpu().double().numpy(); rr = ref[0].numpy()
# is the value somewhere else in the tile?
d = np.abs(yy[:, :, None] - 0)  # placeholder
for dr in range(-8, 9):
    for dc in (-4, -3, -2, -1, 0, 1, 2, 3, 4, 32, -32):
        rr2, cc2 = r + dr, c + dc
        if 0 <= rr2 < m and 0 <= cc2 < t and abs(yy[r, c] - rr[rr2, cc2]) < 1e-4:
            print("y[r,c] equals ref at offset", dr, dc)
print("---- stats")
yy = y[..., :t].cpu().double()
st = st.cpu().numpy()
print("stats sum", st.sum(1)[:, 0], "y sum", yy.sum((1, 2)).numpy())
# per tile / wave expectations: part = (mt*tiles_t + tile)*4 + hw, hw = wm*2+wt: rows wm*128.., cols tile*128 + wt*64 ..
for nn in range(n):
    for tile in range(4):
        for hw in range(4):
            wm, wt = hw >> 1, hw & 1
            c0 = tile * 128 + wt * 64
            exp = yy[nn, wm * 128:wm * 128 + 128, c0:min(c0 + 64, t)].sum().item()
            got = st[nn, tile * 4 + hw, 0]
            if abs(exp - got) > 1e-2 * max(1, abs(exp)):
                print("utt", nn, "tile", tile, "wave", hw, "expected", exp, "got", got)
