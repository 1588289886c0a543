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
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import cases  # noqa: E402
from detweights import det_state_dict  # noqa: E402
import puresound_amd.nnet as PA  # noqa: E402
from puresound_amd import _abi  # noqa: E402


def cfg4(args):
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"bench_recurrent.py --gpus {args.gpus} but WORLD_SIZE = {world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus} (one rank per GPU)")
    if world > 1:
        return cfg4_data_parallel(args, world)
    import bench_configs as BC
    dev = "cuda:0"
    r = BC.cfg4(dev, steps=args.steps, warmup=args.warmup, batch=args.batch, gemm=args.gemm)
    line = {"config": "cfg4 DPRNN(128,64,128,6 blocks,K=20,causal) fp32 storage", "input_projection_gemm": args.gemm,
            "batch": args.batch, "ms_per_forward": r["ms"], "samples_per_s": r["samples_s"],
            "us_per_serial_step": r["us_per_serial_step"], "hipgraph_ms_per_forward": r["hipgraph_ms"],
            "hipgraph_samples_per_s": args.batch * 64000 / r["hipgraph_ms"] * 1e3}
    if args.profile:
        import ctypes as C
        lib = _abi.lib()
        model = BC.cfg4_model(dev, args.gemm)
        noisy = BC._waves(args.batch, 1234, dev)
        model.hip_streams = 1
        model.inference(noisy)
        lib.ps_profile_enable(1)
        model.inference(noisy)
        torch.cuda.synchronize()
        fam = {}
        for k in ("conv1x1", "lstm", "chan_layernorm", "proj_layernorm", "free_encode", "free_decode", "film_apply"):
            ms_k, cnt = C.c_double(), C.c_int()
            lib.ps_profile_read(k.encode(), C.byref(ms_k), C.byref(cnt))
            fam[k] = [round(ms_k.value, 4), cnt.value]
        lib.ps_profile_enable(0)
        line["kernel_ms_per_forward"] = fam
    print(json.dumps(line))


def cfg4_data_parallel(args, world):
    """BASELINE config 4 as north_star states it: batch = 32 * N x 4 s, data-parallel over N GPUs of one node, one rank per
    GPU (torch.distributed.run), utterances sharded with no data-path collective, ONE all-gather of the [32, L] output
    waveforms per step (puresound_amd.batch_shard.sharded_inference -- what replaces the reference's nn.DataParallel,
    task/base.py:226-229).  Weak scaling: per-GPU work is fixed; the time is the max over ranks between barriers."""
    import torch.distributed as dist
    import bench_configs as BC
    from puresound_amd.batch_shard import OverlappedGather, shard_bounds
    rank, local = int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0"))
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    dist.init_process_group("nccl", device_id=dev)
    model = BC.cfg4_model(dev, args.gemm)
    total_b = args.batch * world
    noisy = BC._waves(total_b, 1234, dev)  # every rank holds the synthetic batch and runs its own contiguous share

    lo, hi = shard_bounds(total_b, world, rank)
    overlap = OverlappedGather(total_b)   # step i's all-gather runs under step i+1's kernels (as bench.py --gpus N)

    def step():
        return overlap.submit(model.inference(noisy[lo:hi]))

    def fence():
        overlap.flush()
        torch.cuda.synchronize(dev)
        dist.barrier()
        torch.cuda.synchronize(dev)
    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    mine = torch.tensor([time.perf_counter() - t0], device=dev, dtype=torch.float64)
    every = torch.empty(world, device=dev, dtype=torch.float64)
    dist.all_gather_into_tensor(every, mine)
    elapsed = float(every.max().item())
    if rank == 0:
        print(json.dumps({"config": "cfg4 DPRNN(128,64,128,6 blocks,K=20,causal), data-parallel", "n_gpus": world,
                          "global_batch": total_b, "scaling": "weak", "input_projection_gemm": args.gemm,
                          "ms_per_step": elapsed / args.steps * 1e3, "samples_per_s": total_b * 64000 * args.steps / elapsed,
                          "distributed": {"world_size": dist.get_world_size(), "backend": dist.get_backend(),
                                          "collective": "all_gather_into_tensor of [B/N, L] fp32 per step, asynchronous and double-buffered "
                                                        "(step i's gather under step i+1's kernels; all inside the timed region)",
                                          "ms_per_step_by_rank": [float(v) / args.steps * 1e3 for v in every.tolist()]}}),
              flush=True)
    dist.barrier()
    dist.destroy_process_group()


def dpcrn(args):
    """The real egs/ns model (ns_dpcrn_v0_causal: conv-STFT 512/128 + DPCRN + complex mask + iSTFT), 32 x 4 s."""
    dev = "cuda:0"
    if os.environ.get("PS_FLAGS"):   # experiments: ps_debug_flags for the whole run
        _abi.lib().ps_debug_flags(int(os.environ["PS_FLAGS"], 0))
    model = cases.build(PA.NS, os.environ.get("PS_NS_CASE", "ns_dpcrn_short")).eval()
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
    import bench_configs as BC
    r = BC.cfg5("cuda:0", chunks=args.chunks, streams=args.streams)
    print(json.dumps({"config": "cfg5 demo preset StreamingSkiM(128,256,128,4 blocks,K=150) fp32, one hipGraph per "
                                "320-sample chunk (window shifts, 20 hops, overlap-add, Mem-LSTM update inside)",
                      "streams": args.streams, "chunks": r["chunks"], "chunk_ms_p50": r["p50_ms"],
                      "chunk_ms_p90": r["p90_ms"], "chunk_ms_max": r["max_ms"], "budget_ms": 20.0}))


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--which", default="cfg4,cfg5")
    ap.add_argument("--gpus", type=int, default=1,
                    help="cfg4: data-parallel over N GPUs (starts its own N ranks, or runs under torch.distributed.run): 32 utterances "
                         "per rank, one all-gather of the outputs per step")
    ap.add_argument("--batch", type=int, default=32)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--streams", type=int, default=64)
    ap.add_argument("--chunks", type=int, default=500)
    ap.add_argument("--profile", action="store_true")
    ap.add_argument("--flags", type=int, default=0, help="ps_debug_flags (kernel variant switches)")
    ap.add_argument("--gemm", default="fp32", choices=["fp32", "fp16x2", "bf16x3", "bf16"],
                    help="arithmetic of the LSTM input projections (per module: masker.set_gemm_precision)")
    a = ap.parse_args()
    # `--gpus N` on its own: start the N ranks as child processes before anything here touches a GPU (as bench.py does)
    from puresound_amd import launch
    if launch.needs_self_launch(a.gpus):
        raise SystemExit(launch.self_launch(os.path.abspath(__file__), sys.argv[1:], a.gpus))
    if launch.launch_probe("bench_recurrent.py"):
        raise SystemExit(0)
    if a.flags:
        _abi.lib().ps_debug_flags(a.flags)
    if "cfg4" in a.which:
        cfg4(a)
    if "cfg5" in a.which:
        cfg5(a)
    if "dpcrn" in a.which:
        dpcrn(a)
