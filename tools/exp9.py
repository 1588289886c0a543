"""config 4 with the batch split over 1 / 2 / 4 HIP streams (eager and as one hipGraph)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_configs as BC
dev = "cuda:0"
model = BC.cfg4_model(dev)
noisy = BC._waves(32, 1234, dev)
from puresound_amd.graphs import GraphedInference
for lanes in (1, 2, 4):
    model.hip_streams = lanes
    ms, _ = BC._timed(lambda: model.inference(noisy), 20, 5)
    fast = GraphedInference(model)
    msg, _ = BC._timed(lambda: fast(noisy), 20, 5)
    print(f"hip_streams {lanes}: eager {ms:.3f} ms, hipGraph {msg:.3f} ms", flush=True)
