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
BF16_MFMA_PEAK_TFLOPS = 2500.0  # same guide, dense bf16 matrix peak; the 3-way split spends 6 bf16 products per fp32 one
SPLIT_PRODUCTS = 6
# HBM bytes per ps_conv1x1_bf16_f32 launch (planes = 3) at 16 utterances, from the PMC passes committed as
# profiles/r01_pmc_hbm_traffic_bf16x3.txt (same recipe as below): in_conv + pointwise + out_conv.
PMC_BF16X3_BYTES_PER_LAUNCH = 2 * (207.9e6 + 137.9e6 + 348.9e6) / 3  # algorithmic: 2 * (207.6 + 138.4 + 346.0) / 3 MB
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
    ap.add_argument("--streams", type=int, default=None,
                    help="HIP streams the batch is split over inside one GPU (default: the model's own default)")
    ap.add_argument("--gemm", default="bf16x3", choices=["fp32", "bf16x3", "bf16"],
                    help="arithmetic of the 1x1-conv GEMMs.  bf16x3 (default): every fp32 operand split into three "
                         "bf16 terms, six products on the bf16 MFMA pipe, fp32 accumulation -- the result carries "
                         "the same error against the reference as the exact fp32 MFMA path (9.8e-7 vs 1.13e-6 "
                         "max-rel on the config-2 golden vector; tests/test_hip_parity.py).  fp32: v_mfma_f32_* on "
                         "fp32 operands.  bf16: operands rounded to bf16 (NOT an fp32 result; experiment only).")
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
    model.masker.set_gemm_precision(args.gemm)
    if args.streams is not None:
        model.hip_streams = args.streams

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
        "dtype": {"fp32": "f32",
                  "bf16x3": "f32 (GEMM products issued as 3xbf16 split terms, fp32 accumulate; result error equals the "
                            "exact-fp32 MFMA path's)",
                  "bf16": "f32 storage/accumulate, bf16 products (EXPERIMENT, not an fp32 result)"}[args.gemm],
        "gemm": args.gemm,
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
        for fam in ("dwconv", "free_encode", "free_decode"):
            ms, cnt = ctypes.c_double(), ctypes.c_int()
            _abi.check(lib.ps_profile_read(fam.encode(), ctypes.byref(ms), ctypes.byref(cnt)), "ps_profile_read")
            fams[fam] = (ms.value, cnt.value)
        flops, launches = conv_flops_per_forward(model, B_PER_GPU, t)
        fam = "conv1x1" if args.gemm == "fp32" else "conv1x1_bf16"
        ms, cnt = ctypes.c_double(), ctypes.c_int()
        _abi.check(lib.ps_profile_read(fam.encode(), ctypes.byref(ms), ctypes.byref(cnt)), "ps_profile_read")
        conv_ms, conv_cnt = ms.value, cnt.value
        fams[fam] = (conv_ms, conv_cnt)
        assert conv_cnt == launches * args.steps, (conv_cnt, launches, args.steps)
        avg_ms = conv_ms / conv_cnt
        achieved = (flops / launches) / (avg_ms * 1e-3) / 1e12
        common = {"bound": "mfma", "achieved": achieved, "unit": "TFLOP/s", "avg_launch_ms": avg_ms,
                  "note": "kernel durations from a single-stream pass; value/ms_per_step from the product path "
                          "(2 sub-batch streams)",
                  "flop_per_launch": flops / launches, "launches_per_step": launches,
                  "kernel_ms_per_step": {k: v[0] / args.steps for k, v in fams.items()}}
        if args.gemm == "fp32":
            result["roofline"] = dict(common, peak=F32_MFMA_PEAK_TFLOPS, frac=achieved / F32_MFMA_PEAK_TFLOPS,
                                      traffic=PMC_CONV1X1_BYTES_PER_LAUNCH, kernel="ps::conv1x1_* (ps_conv1x1_f32)",
                                      traffic_note="bytes per launch from the committed PMC passes "
                                                   "(profiles/r01_pmc_hbm_traffic.txt), not re-measured in this run")
        else:
            planes_products = SPLIT_PRODUCTS if args.gemm == "bf16x3" else 1
            peak = BF16_MFMA_PEAK_TFLOPS / planes_products
            result["roofline"] = dict(
                common, peak=peak, frac=achieved / peak, traffic=PMC_BF16X3_BYTES_PER_LAUNCH if args.gemm == "bf16x3" else None,
                kernel="ps::conv1x1_bf16_pp_kernel (ps_conv1x1_bf16_f32)",
                peak_note=f"algorithmic fp32 FLOP (2*M*K*T*N per launch) against the dense bf16 MFMA peak "
                          f"{BF16_MFMA_PEAK_TFLOPS:.0f} TFLOP/s / {planes_products} bf16 products per multiply-add; "
                          f"the same FLOP against the fp32 MFMA peak {F32_MFMA_PEAK_TFLOPS} TFLOP/s = "
                          f"{achieved / F32_MFMA_PEAK_TFLOPS:.2f}",
                traffic_note="bytes per launch from the committed PMC passes "
                             "(profiles/r01_pmc_hbm_traffic_bf16x3.txt), not re-measured in this run")
    if rank == 0 and world == 1 and args.gemm == "bf16x3" and not args.no_roofline:
        # the exact-fp32 MFMA path (v_mfma_f32_32x32x2_f32 on fp32 operands) timed beside it, same inputs, same K steps
        model.masker.set_gemm_precision("fp32")
        for _ in range(max(2, args.warmup)):
            model.inference(noisy)
        torch.cuda.synchronize(dev)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            model.inference(noisy)
        torch.cuda.synchronize(dev)
        dt = time.perf_counter() - t0
        model.masker.set_gemm_precision(args.gemm)
        result["fp32_mfma_path"] = {"value": B_PER_GPU * L * args.steps / dt, "unit": "samples/s",
                                    "ms_per_step": dt / args.steps * 1e3,
                                    "note": "python bench.py --gemm fp32: the same step with v_mfma_f32 on fp32 operands"}
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
