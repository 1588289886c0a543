#!/usr/bin/env python3
"""Headline benchmark: audio samples/sec (16 kHz) of the ns Conv-TasNet forward path, batch = 32 x 4 s fp32
per GPU (BASELINE.json configs[1]), whole-job aggregate over N GPUs (weak scaling: every rank owns its own
32 utterances, one RCCL all-gather reassembles the output waveforms inside the timed step).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

A step = SoTaskWrapModule.inference on one resident batch: encoder -> 24 TCN blocks -> mask -> decoder -> clamp.
Rank 0 prints ONE JSON line (see DESIGN.md "Measurement" for the roofline / cpu_baseline definitions).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))

import torch  # noqa: E402

B_PER_GPU, L, SR = 32, 64000, 16000
F32_MFMA_PEAK_TFLOPS = 157.3  # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
# HBM bytes per ps_conv1x1_f32 launch at 32 utterances, averaged over the three GEMM shapes, from the PMC
# passes committed as profiles/r01_pmc_hbm_traffic.txt (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate
# runs, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950): (207.4 + 149.4 + 388.4) / 3 MB per
# 16-utterance launch.  Algorithmic bytes of the same launches: (207.6 + 138.4 + 346.0) / 3 MB.
PMC_CONV1X1_BYTES_PER_LAUNCH = 2 * (207.4e6 + 149.4e6 + 388.4e6) / 3


def build_model(dev):
    import cases
    import puresound_amd.nnet as PA
    torch.manual_seed(0)
    model = cases.build(PA.NS, "cfg2_full").eval()  # random-init weights of the named architecture
    return model.to(dev)


def conv_flops_per_forward(model, n, t):
    """Algorithmic FLOPs of the 1x1-conv GEMM launches of one forward: 2*M*K*T*N per launch."""
    c, h = model.masker.input_dim, model.masker.tcn_dim
    blocks = model.masker.repeat_tcn * model.masker.per_tcn_stack
    per_block = 2.0 * n * t * (c * h + h * h + h * c)
    return blocks * per_block, blocks * 3


def host_cores():
    """CPU threads this process may actually use (affinity mask and cgroup quota, not the host's total)."""
    n = len(os.sched_getaffinity(0))
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except (OSError, ValueError):
        pass
    return n


def cpu_baseline(seconds_budget=25.0):
    """The CPU oracle (a restatement of the reference's PyTorch CPU path, pinned to it by tests/golden)
    timed on this host's cores on a bounded sample of the same workload."""
    import cases
    from detweights import det_state_dict, det_wave
    from oracle import separator_oracle as O
    import puresound_amd.nnet as PA
    model = cases.build(PA.NS, "cfg2_full")
    sd = {k: v.float() for k, v in det_state_dict(model).items()}
    cfg = cases.oracle_cfg("cfg2_full")
    nb = 4
    torch.set_num_threads(host_cores())
    x = det_wave(1234, nb, L)
    with torch.no_grad():
        O.inference(x[:1], sd, cfg)  # warm-up (thread pools, allocator)
        t0 = time.perf_counter()
        reps = 0
        while True:
            O.inference(x, sd, cfg)
            reps += 1
            if time.perf_counter() - t0 > seconds_budget * 0.6 or reps >= 3:
                break
        dt = (time.perf_counter() - t0) / reps
    return {"value": nb * L / dt, "unit": "samples/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{reps} x batch {nb} x 4 s of the same config (oracle/separator_oracle.py, fp32, torch CPU)"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--gemm", default="fp32", choices=["fp32", "bf16x3", "bf16"],
                    help="matrix-pipe arithmetic of the 1x1 convs; the BASELINE metric is fp32 (default). The other "
                         "modes are experiments and are labelled as such in the output line")
    args = ap.parse_args()

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N > 1")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from puresound_amd import _abi
    from puresound_amd.batch_shard import gather_utterances
    lib = _abi.lib()  # no HIP extension, no benchmark
    model = build_model(dev)
    if args.gemm != "fp32":
        model.masker.set_gemm_precision(args.gemm)
        args.no_roofline = True  # the roofline object describes the fp32 MFMA kernel only

    g = torch.Generator().manual_seed(1234 + rank)
    noisy = ((torch.rand(B_PER_GPU, L, generator=g) * 2 - 1) * 0.5).to(dev)  # synthetic 16 kHz waveforms
    total_b = B_PER_GPU * world

    def step():
        out = model.inference(noisy)
        if world > 1:
            out = gather_utterances(out, total_b)
        return out

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    ms_per_step = elapsed / args.steps * 1e3
    value = total_b * L * args.steps / elapsed

    result = {
        "metric": "audio samples/sec (16 kHz) on ns Conv-TasNet, batch=32x4s per GPU",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32" if args.gemm == "fp32" else ("f32 storage/accumulate, " + args.gemm + " products (EXPERIMENT)"),
        "data": "synthetic",
        "config": {"workload": "egs/ns Conv-TasNet, learned-conv encoder (32/16/512), R=3 X=8 H=256, "
                               "batch=32x4 s fp32 per GPU (BASELINE configs[1])",
                   "global_batch": total_b, "samples_per_utt": L, "parallelism": f"dp{world}",
                   "hip_streams_per_gpu": int(getattr(model, "hip_streams", 2)),
                   "x_realtime": value / SR},
    }

    if rank == 0 and not args.no_roofline:
        # second pass over the same K steps with the library's per-launch hipEvents switched on.  The
        # sub-batch stream overlap is switched off for this pass: with two launches sharing the chip an
        # event pair would time the overlap, not the kernel.
        t = (L - 32) // 16 + 1
        streams_kept = getattr(model, "hip_streams", 2)
        model.hip_streams = 1
        model.inference(noisy)
        torch.cuda.synchronize(dev)
        lib.ps_profile_enable(1)
        for _ in range(args.steps):
            model.inference(noisy)
        torch.cuda.synchronize(dev)
        lib.ps_profile_enable(0)
        model.hip_streams = streams_kept
        import ctypes
        fams = {}
        for fam in ("conv1x1", "dwconv", "free_encode", "free_decode"):
            ms, cnt = ctypes.c_double(), ctypes.c_int()
            _abi.check(lib.ps_profile_read(fam.encode(), ctypes.byref(ms), ctypes.byref(cnt)), "ps_profile_read")
            fams[fam] = (ms.value, cnt.value)
        flops, launches = conv_flops_per_forward(model, B_PER_GPU, t)
        conv_ms, conv_cnt = fams["conv1x1"]
        assert conv_cnt == launches * args.steps, (conv_cnt, launches, args.steps)
        avg_ms = conv_ms / conv_cnt
        achieved = (flops / launches) / (avg_ms * 1e-3) / 1e12
        result["roofline"] = {"bound": "mfma", "achieved": achieved, "peak": F32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
                              "frac": achieved / F32_MFMA_PEAK_TFLOPS, "traffic": PMC_CONV1X1_BYTES_PER_LAUNCH,
                              "traffic_note": "bytes per launch from the committed PMC passes "
                                              "(profiles/r01_pmc_hbm_traffic.txt), not re-measured in this run",
                              "kernel": "ps::conv1x1_* (ps_conv1x1_f32)", "avg_launch_ms": avg_ms,
                              "note": "kernel durations from a single-stream pass; value/ms_per_step from the "
                                      "product path (2 sub-batch streams)",
                              "flop_per_launch": flops / launches, "launches_per_step": launches,
                              "kernel_ms_per_step": {k: v[0] / args.steps for k, v in fams.items()}}
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
