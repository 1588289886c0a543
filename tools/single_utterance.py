"""Latency of ONE 4 s utterance through the config-2 model (north star: >= 30 x real time per utterance)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import json
import numpy as np
import torch
import bench
from puresound_amd.graphs import GraphedInference

dev = torch.device("cuda:0")
model = bench.build_model(dev)
g = torch.Generator().manual_seed(1234)
for gemm in ("bf16x3", "fp16x2"):
    model.masker.set_gemm_precision(gemm)
    for n in (1, 4):
        noisy = ((torch.rand(n, bench.L, generator=g) * 2 - 1) * 0.5).to(dev)
        for _ in range(10):
            model.inference(noisy)
        torch.cuda.synchronize()
        lat = []
        for _ in range(50):
            t0 = time.perf_counter()
            model.inference(noisy)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
        lat = np.array(lat)
        print(json.dumps({"config": "config 2 model, batch %d x 4 s" % n, "gemm": gemm, "launch": "eager",
                          "ms_p50": float(np.percentile(lat, 50)), "ms_p90": float(np.percentile(lat, 90)),
                          "x_realtime_p50": n * 4000.0 / float(np.percentile(lat, 50))}), flush=True)
        fast = GraphedInference(model)
        for _ in range(5):
            fast(noisy)
        torch.cuda.synchronize()
        lat = []
        for _ in range(50):
            t0 = time.perf_counter()
            fast(noisy)
            torch.cuda.synchronize()
            lat.append((time.perf_counter() - t0) * 1e3)
        lat = np.array(lat)
        print(json.dumps({"config": "config 2 model, batch %d x 4 s" % n, "gemm": gemm, "launch": "hipGraph replay",
                          "ms_p50": float(np.percentile(lat, 50)), "ms_p90": float(np.percentile(lat, 90)),
                          "x_realtime_p50": n * 4000.0 / float(np.percentile(lat, 50))}), flush=True)
