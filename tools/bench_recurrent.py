"""Timing of the recurrent configs (not the bench.py headline): BASELINE config 4 (DPRNN, 32 x 4 s per GPU) as
ms/forward + samples/s, config 5 (demo preset, 64 streams, 320-sample chunks) as p50/p90 chunk latency.
Prints one JSON line per config.  --profile brackets every kernel launch with events (ps_profile_*)."""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
import cases  # noqa: E402
from detweights import det_state_dict  # noqa: E402
import puresound_amd.nnet as PA  # noqa: E402
from puresound_amd import _abi  # noqa: E402


def cfg4(args):
    dev = "cuda:0"
    model = cases.build(PA.NS, "cfg4_short").eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    model.masker.set_gemm_precision(args.gemm)
    if args.hip_streams:
        model.hip_streams = args.hip_streams
    g = torch.Generator().manual_seed(1234)
    noisy = ((torch.rand(args.batch, 64000, generator=g) * 2 - 1) * 0.5).to(dev)
    for _ in range(args.warmup):
        model.inference(noisy)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.inference(noisy)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    line = {"config": "cfg4 DPRNN(128,64,128,6 blocks,K=20,causal) fp32 storage", "input_projection_gemm": args.gemm,
            "batch": args.batch, "ms_per_forward": ms, "hip_streams": int(getattr(model, "hip_streams", 2)),
            "samples_per_s": args.batch * 64000 / ms * 1e3}
    # the same forward replayed as one hipGraph (45 short kernels: the eager path leaves gaps between them)
    from puresound_amd.graphs import GraphedInference
    fast = GraphedInference(model)
    for _ in range(3):
        fast(noisy)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        fast(noisy)
    torch.cuda.synchronize()
    msg = (time.perf_counter() - t0) / args.steps * 1e3
    line["hipgraph_ms_per_forward"] = msg
    line["hipgraph_samples_per_s"] = args.batch * 64000 / msg * 1e3
    if args.profile:
        import ctypes as C
        lib = _abi.lib()
        model.hip_streams = 1
        lib.ps_profile_enable(1)
        model.inference(noisy)
        torch.cuda.synchronize()
        fam = {}
        for k in ("conv1x1", "lstm", "chan_layernorm", "free_encode", "free_decode", "film_apply"):
            ms_k, cnt = C.c_double(), C.c_int()
            lib.ps_profile_read(k.encode(), C.byref(ms_k), C.byref(cnt))
            fam[k] = [round(ms_k.value, 4), cnt.value]
        lib.ps_profile_enable(0)
        line["kernel_ms_per_forward"] = fam
    print(json.dumps(line))


def dpcrn(args):
    """The real egs/ns model (ns_dpcrn_v0_causal: conv-STFT 512/128 + DPCRN + complex mask + iSTFT), 32 x 4 s."""
    dev = "cuda:0"
    model = cases.build(PA.NS, "ns_dpcrn_short").eval()
    model.load_state_dict(det_state_dict(model))
    model.to(dev)
    model.masker.set_gemm_precision(args.gemm)
    g = torch.Generator().manual_seed(1234)
    noisy = ((torch.rand(args.batch, 64000, generator=g) * 2 - 1) * 0.5).to(dev)
    for _ in range(args.warmup):
        model.inference(noisy)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        model.inference(noisy)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / args.steps * 1e3
    line = {"config": "ns_dpcrn_v0_causal (STFT 512/128 + DPCRN(1,32,32,32,64,128; H=128)) fp32 storage",
            "input_projection_gemm": args.gemm, "batch": args.batch,
            "ms_per_forward": ms, "samples_per_s": args.batch * 64000 / ms * 1e3}
    if args.profile:
        import ctypes as C
        lib = _abi.lib()
        lib.ps_profile_enable(1)
        model.inference(noisy)
        torch.cuda.synchronize()
        fam = {}
        for k in ("conv1x1", "unfold2d", "activation", "lstm", "proj_layernorm", "frame", "istft_ola", "complex_mask"):
            ms_k, cnt = C.c_double(), C.c_int()
            lib.ps_profile_read(k.encode(), C.byref(ms_k), C.byref(cnt))
            fam[k] = [round(ms_k.value, 4), cnt.value]
        lib.ps_profile_enable(0)
        line["kernel_ms_per_forward"] = fam
    print(json.dumps(line))


def cfg5(args):
    from puresound_amd.streaming.demo import DemoTseNet
    dev = "cuda:0"
    net = DemoTseNet().eval()
    net.load_state_dict(det_state_dict(net))
    net.to(dev)
    b = args.streams
    net.init_streams(b)
    g = torch.Generator().manual_seed(1236)
    embed = torch.rand(b, 192, generator=g).to(dev)
    wav = ((torch.rand(b, 320 * 8, generator=g) * 2 - 1) * 0.5).to(dev)
    lat = {"hop_graphs": [], "chunk_graph": []}
    for mode in ("hop_graphs", "chunk_graph"):   # 20 replays of the hop graph (round 1) | one graph per chunk, OLA included
        net.init_streams(b)
        pre = None
        for i in range(args.chunks + 10):
            chunk = wav[:, (i % 8) * 320:(i % 8 + 1) * 320]
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            if mode == "hop_graphs":
                for j in range(20):
                    net.streaming_inference(chunk[:, j * 16:(j + 1) * 16], embed)
            else:
                out = net.streaming_inference_chunk(chunk, embed, pre)
                pre = out[:, -16:]
            torch.cuda.synchronize()
            if i >= 10:
                lat[mode].append((time.perf_counter() - t0) * 1e3)
    hop = np.array(lat["hop_graphs"])
    lat = np.array(lat["chunk_graph"])
    print(json.dumps({"config": "cfg5 demo preset StreamingSkiM(128,256,128,4 blocks,K=150) fp32, one hipGraph per "
                                "320-sample chunk (window shifts, 20 hops, overlap-add, Mem-LSTM update inside)",
                      "streams": b, "hop_graph_chunk_ms_p50": float(np.percentile(hop, 50)),
                      "chunks": len(lat), "chunk_ms_p50": float(np.percentile(lat, 50)),
                      "chunk_ms_p90": float(np.percentile(lat, 90)), "chunk_ms_max": float(lat.max()),
                      "budget_ms": 20.0}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="cfg4,cfg5")
    ap.add_argument("--hip-streams", type=int, default=0, help="cfg4: HIP streams the batch is split over (0 = the model's default)")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--chunks", type=int, default=500)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="ps_debug_flags (kernel variant switches)")
    ap.add_argument("--gemm", default="fp32", choices=["fp32", "bf16x3", "bf16"],
                    help="arithmetic of the LSTM input projections (puresound_amd.nnet._plans.set_recurrent_gemm_precision)")
    a = ap.parse_args()
    if a.flags:
        _abi.lib().ps_debug_flags(a.flags)
    from puresound_amd.nnet import _plans
    if "cfg4" in a.which:
        cfg4(a)
    if "cfg5" in a.which:
        cfg5(a)
    if "dpcrn" in a.which:
        dpcrn(a)
