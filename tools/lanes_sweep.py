"""Step time of the headline configuration as a function of the number of sub-batch HIP streams."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
import torch
import cases
import puresound_amd.nnet as PA
dev = "cuda:0"
torch.manual_seed(0)
model = cases.build(PA.NS, "cfg2_full").eval().to(dev)
g = torch.Generator().manual_seed(1234)
noisy = ((torch.rand(32, 64000, generator=g) * 2 - 1) * 0.5).to(dev)
for lanes in (1, 2, 3, 4, 2):
    model.hip_streams = lanes
    for _ in range(3):
        model.inference(noisy)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        model.inference(noisy)
    torch.cuda.synchronize()
    print(f"hip_streams={lanes}: {(time.perf_counter() - t0) / 10 * 1e3:.2f} ms/step", flush=True)
