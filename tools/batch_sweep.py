"""Config-2 forward time against the batch size, per GEMM kernel choice (ps_debug_flags) -- run on the GPU box."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from puresound_amd import _abi

dev = torch.device("cuda:0")
model = bench.build_model(dev)
model.masker.set_gemm_precision(sys.argv[1] if len(sys.argv) > 1 else "bf16x3")
g = torch.Generator().manual_seed(1234)
for n in (1, 2, 4, 8, 12, 16, 24, 32):
    noisy = ((torch.rand(n, bench.L, generator=g) * 2 - 1) * 0.5).to(dev)
    line = [f"batch {n:2d}:"]
    for name, flags in (("auto", 0), ("simple", 1 << 27), ("simple-wide", (1 << 27) | (1 << 29)), ("pingpong", 1 << 28)):
        _abi.lib().ps_debug_flags(flags)
        for streams in (1, 2):
            model.hip_streams = streams
            for _ in range(5):
                model.inference(noisy)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(15):
                model.inference(noisy)
            torch.cuda.synchronize()
            line.append(f"{name}/s{streams} {(time.perf_counter() - t0) / 15 * 1e3:6.2f}")
    _abi.lib().ps_debug_flags(0)
    print("  ".join(line), flush=True)
