import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import bench_configs as BC
from puresound_amd.graphs import GraphedInference
dev = "cuda:0"
noisy = BC._waves(32, 1234, dev)
for lanes in (1, 2):
    for gemm in ("fp32", "fp16x2"):
        os.environ["PS_CFG4_STREAMS"] = str(lanes)
        model = BC.cfg4_model(dev, gemm)
        ms, _ = BC._timed(lambda: model.inference(noisy), 20, 5)
        fast = GraphedInference(model)
        msg, _ = BC._timed(lambda: fast(noisy), 20, 5)
        print(f"cfg4 streams {lanes} {gemm}: eager {ms:.3f} ms, hipGraph {msg:.3f} ms", flush=True)
