"""config 3 by kernel family (the library's launch timers)."""
import ctypes, os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_configs as BC
from puresound_amd import _abi
dev = "cuda:0"; lib = _abi.lib()
model = BC._build("cfg3_short", dev)
model.hip_streams = 1
noisy, enroll = BC._waves(32, 1234, dev), BC._waves(32, 1235, dev)
for _ in range(3): model.inference(noisy, enroll)
torch.cuda.synchronize()
ms, _ = BC._timed(lambda: model.inference(noisy, enroll), 10, 2)
lib.ps_profile_enable(1)
for _ in range(5): model.inference(noisy, enroll)
torch.cuda.synchronize(); lib.ps_profile_enable(0)
tot = 0
for fam in ("conv1x1_bf16", "conv1x1", "dwconv", "free_encode", "free_decode", "absmax", "attn_stats_pool", "embed_bias", "chan_layernorm", "pad_rows", "unpad_rows", "row_stats", "norm_activation"):
    v, c = ctypes.c_double(), ctypes.c_int()
    lib.ps_profile_read(fam.encode(), ctypes.byref(v), ctypes.byref(c))
    if c.value:
        print(f"  {fam:18s} {v.value / 5:.3f} ms/forward ({v.value / c.value * 1e3:.1f} us x {c.value // 5})"); tot += v.value / 5
print(f"cfg3 one stream: {ms:.3f} ms per forward, timed kernels {tot:.3f} ms")
