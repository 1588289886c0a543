"""Does a smaller working set (hidden maps resident in the 256 MB memory-side cache) beat the launch efficiency of the
full batch?  32 utterances x 4 s as 32/B sequential sub-batches of B, eager, one stream (run on the GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import bench
dev = torch.device("cuda:0")
model = bench.build_model(dev)
model.hip_streams = 1
from puresound_amd import _abi
_abi.lib().ps_debug_flags(int(os.environ.get("PS_FLAGS", "0"), 0))
x = ((torch.rand(32, 64000) * 2 - 1) * 0.5).to(dev)
for gemm in ("fp16x2",):
    model.masker.set_gemm_precision(gemm)
    for b in (32, 16, 8):
        chunks = list(x.split(b))
        for _ in range(2):
            for c in chunks:
                model.inference(c)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = 5
        for _ in range(reps):
            for c in chunks:
                model.inference(c)
        torch.cuda.synchronize()
        print(f"{gemm} sub-batch {b:2d}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per 32 utterances", flush=True)
