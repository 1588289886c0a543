"""config 2's step eager vs replayed as one hipGraph (puresound_amd.graphs.GraphedInference), 32 x 4 s, one stream"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import bench
from puresound_amd.graphs import GraphedInference
dev = torch.device("cuda:0")
model = bench.build_model(dev)
model.hip_streams = 1
x = ((torch.rand(32, 64000) * 2 - 1) * 0.5).to(dev)
def timed(fn, reps=20):
    for _ in range(5): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps * 1e3
for rnd in range(2):
    e = timed(lambda: model.inference(x))
    g = GraphedInference(model)
    gr = timed(lambda: g(x))
    y0, y1 = model.inference(x), g(x)
    print(f"eager {e:.3f} ms   hipGraph {gr:.3f} ms   identical {bool(torch.equal(y0, y1))}", flush=True)
