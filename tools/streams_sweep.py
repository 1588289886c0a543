"""Config-2 step time against the number of sub-batch HIP streams, per GEMM arithmetic (run on the GPU box)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
g = torch.Generator().manual_seed(1234)
noisy = ((torch.rand(bench.B_PER_GPU, bench.L, generator=g) * 2 - 1) * 0.5).to(dev)
for gemm in sys.argv[1:] or ["fp32", "bf16x3", "bf16"]:
    model = bench.build_model(dev)
    if gemm != "fp32":
        model.masker.set_gemm_precision(gemm)
    from puresound_amd import _abi
    for streams, cap in ((1, 0), (2, 0), (2, 128), (2, 160), (4, 64), (4, 128), (3, 96)):
        _abi.lib().ps_debug_flags(cap << 8)
        model.hip_streams = streams
        for _ in range(5):
            model.inference(noisy)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            model.inference(noisy)
        torch.cuda.synchronize()
        print(f"{gemm:7s} hip_streams={streams} grid cap {cap}: {(time.perf_counter() - t0) / 20 * 1e3:.2f} ms/step", flush=True)
