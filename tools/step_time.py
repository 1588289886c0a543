"""The benchmark's step (config 2, 32 x 4 s, fp16x2 by default) with the library's per-launch hipEvents: ms per step and
average launch of each kernel family.  PURESOUND_HIP_LIB=tools/_variants/X.so python tools/step_time.py [gemm] [steps]"""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from puresound_amd import _abi

gemm = sys.argv[1] if len(sys.argv) > 1 else "fp16x2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda:0")
lib = _abi.lib()
model = bench.build_model(dev)
model.masker.set_gemm_precision(gemm)
model.hip_streams = int(os.environ.get("PS_STREAMS", "1"))
lib.ps_debug_flags((int(os.environ.get("PS_CAP", "0")) << 8) | int(os.environ.get("PS_FLAGS", "0"), 0))
g = torch.Generator().manual_seed(1234)
noisy = ((torch.rand(bench.B_PER_GPU, bench.L, generator=g) * 2 - 1) * 0.5).to(dev)
for _ in range(5):
    model.inference(noisy)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(steps):
    model.inference(noisy)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / steps * 1e3
lib.ps_profile_enable(1)
for _ in range(steps):
    model.inference(noisy)
torch.cuda.synchronize()
lib.ps_profile_enable(0)
out = []
for fam in ("conv1x1_bf16", "conv1x1", "dwconv", "free_encode", "free_decode", "absmax"):
    v, c = ctypes.c_double(), ctypes.c_int()
    lib.ps_profile_read(fam.encode(), ctypes.byref(v), ctypes.byref(c))
    if c.value:
        out.append(f"{fam} {v.value / steps:.3f} ms/step ({v.value / c.value * 1e3:.1f} us x {c.value // steps})")
if "--stamps" in sys.argv or os.environ.get("PS_STAMPS"):
    import numpy as np
    buf = torch.zeros(512 * 6, dtype=torch.int64, device=dev)
    lib.ps_debug_buffer(buf.data_ptr())
    model.inference(noisy)
    torch.cuda.synchronize(); lib.ps_debug_buffer(None)
    q = buf.cpu().numpy().reshape(512, 6)[0::2]
    print(f"last GEMM launch of the step (out_conv): {np.median(q[:, 0]):.0f} cycles in {np.median(q[:, 2]) / 100:.1f} us = "
          f"{np.median(q[:, 0] / np.maximum(q[:, 2], 1)) / 10:.2f} GHz; {np.median(q[:, 0] / np.maximum(q[:, 5], 1)):.0f} cycles/step")
tag = os.path.basename(os.environ.get("PURESOUND_HIP_LIB", "default"))
print(f"{tag} [{gemm}] streams={model.hip_streams} cap={os.environ.get('PS_CAP', 0)} flags={os.environ.get('PS_FLAGS', 0)} {ms:.3f} ms/step | " + " | ".join(out), flush=True)
