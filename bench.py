#!/usr/bin/env python3
"""Headline benchmark: audio samples/sec (16 kHz) of the ns Conv-TasNet forward path, batch = 32 x 4 s fp32
per GPU (BASELINE.json configs[1]), whole-job aggregate over N GPUs (weak scaling: every rank owns its own
32 utterances, one RCCL all-gather reassembles the output waveforms inside the timed step).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python bench.py --gpus N --steps K --warmup W          (starts its own N ranks: puresound_amd/launch.py)
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
SPLIT_PRODUCTS = {"bf16x3": 6, "fp16x2": 3, "bf16": 1}  # 16-bit MFMA products per fp32 multiply-add
HBM_PEAK_GBPS = 8000.0  # same guide: HBM3E, ~8 TB/s
# roofline.traffic (HBM bytes per launch of the dominant kernel) cannot be collected inside this run -- PMC counters
# need their own rocprofv3 passes -- so it is READ from the summary of those passes committed under profiles/
# (tools/pmc_summary.py; same command line as this benchmark, FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for
# gfx950) and labelled with the file it came from.  It never enters `frac`.
PMC_FILES = {"fp16x2": "profiles/r04_pmc_hbm_traffic_fp16x2.json", "bf16x3": "profiles/r04_pmc_hbm_traffic_bf16x3.json",
             "fp32": "profiles/r04_pmc_hbm_traffic_fp32.json"}


def pmc_traffic(gemm, prefix):
    """(average HBM bytes per launch over the kernels whose name starts with `prefix`, source file) or (None, None)."""
    path = os.path.join(ROOT, PMC_FILES.get(gemm, ""))
    if not os.path.isfile(path):
        return None, None
    prefixes = (prefix,) if isinstance(prefix, str) else tuple(prefix)
    ks = {k: v for k, v in json.load(open(path))["kernels"].items() if k.startswith(prefixes)}
    n = sum(v["launches"] for v in ks.values())
    if not n:
        return None, None
    return sum(v["total_bytes"] * v["launches"] for v in ks.values()) / n, PMC_FILES[gemm]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


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


def cpu_baseline(seconds_budget=75.0):
    """The CPU oracle (a restatement of the reference's PyTorch CPU path, pinned to it by tests/golden) timed on this
    host's cores as SURVEY 8(d) asks: batches of 1, 4, 8 (medians of up to five runs, one warm-up each) and one run of the
    full batch of 32 on all the cores this process may use, and one utterance on one core.  `value` is the batch-8 median
    (what round 3 reported); round 2's 0.60 M samples/s was the MEAN of <= 3 runs of batch 4 -- the per-sample rate of
    this port depends on the batch (working set against the last-level cache), which is why every batch is in the line.
    Runs are cut short (and the line says how many were made) when they would exceed the budget."""
    import statistics
    import cases
    from detweights import det_state_dict, det_wave
    from oracle import separator_oracle as O
    import puresound_amd.nnet as PA
    model = cases.build(PA.NS, "cfg2_full")
    sd = {k: v.float() for k, v in det_state_dict(model).items()}
    cfg = cases.oracle_cfg("cfg2_full")
    x = det_wave(1234, 32, L)
    cores = host_cores()
    plan = [(cores, 8, 5, 0.30), (cores, 1, 5, 0.08), (cores, 4, 5, 0.14), (1, 1, 5, 0.12), (cores, 32, 1, 0.36)]
    runs = []
    with torch.no_grad():
        torch.set_num_threads(cores)
        O.inference(x[:1, :16000], sd, cfg)  # warm-up (thread pools, allocator)
        for threads, batch, reps, share in plan:
            torch.set_num_threads(threads)
            t_start = time.perf_counter()
            times = []
            if batch < 32:
                O.inference(x[:batch], sd, cfg)  # warm-up at the timed size
            while len(times) < reps and (not times or (time.perf_counter() - t_start) < seconds_budget * share):
                t0 = time.perf_counter()
                O.inference(x[:batch], sd, cfg)
                times.append(time.perf_counter() - t0)
            runs.append({"threads": threads, "batch": batch, "runs": len(times), "samples_s": batch * L / statistics.median(times),
                         "median_s": statistics.median(times)})
    head = runs[0]
    one = [r for r in runs if r["threads"] == 1][0]
    return {"value": head["samples_s"], "unit": "samples/s", "cores": cores, "kind": "port", "cpu_model": cpu_model(),
            "sample": f"median of {head['runs']} timed runs (1 warm-up) of batch 8 x 4 s of the benchmark's configuration "
                      f"(oracle/separator_oracle.py, fp32, torch CPU, {cores} threads); the other batches and the "
                      f"single-thread run are under `runs`",
            "runs": runs,
            "single_thread": {"value": one["samples_s"], "unit": "samples/s", "cores": 1,
                              "sample": f"median of {one['runs']} timed runs of batch 1 x 4 s, torch.set_num_threads(1)"}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-roofline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true",
                    help="skip the timed lines of BASELINE configs 3, 4 and 5 (`other_configs`; they add ~20 s)")
    ap.add_argument("--streams", type=int, default=1,
                    help="HIP streams the batch is split over inside one GPU.  Default 1: every launch covers the whole "
                         "batch, so the timed path, the per-kernel hipEvents and a rocprofv3 kernel trace of this "
                         "command describe the same launches (two streams bring the split GEMM nothing: 13.5 vs 13.7 ms)")
    ap.add_argument("--ragged", action="store_true",
                    help="multi-GPU: uneven shards (rank r owns 32 - r utterances) through "
                         "puresound_amd.batch_shard.sharded_inference's ragged gather instead of equal shards")
    ap.add_argument("--sync-gather", action="store_true",
                    help="multi-GPU: issue each step's all-gather synchronously (the next step waits for it) instead of "
                         "overlapping it with the next step's kernels")
    ap.add_argument("--gemm", default="fp16x2", choices=["fp32", "bf16x3", "fp16x2", "bf16"],
                    help="arithmetic of the 1x1-conv GEMMs; tensors and accumulation are fp32 in every case.  fp16x2 "
                         "(default): every fp32 operand as two fp16 terms, three products on the fp16 MFMA pipe, "
                         "operands scaled into fp16's range by powers of two -- against fp64 products and against the "
                         "reference's golden vectors its error is that of the exact fp32 MFMA path (config 2, pre-clamp "
                         "max-rel: 9.8e-7 vs 1.03e-6; tests/test_fp16x2.py).  bf16x3: three bf16 terms, six products, "
                         "fp32-accurate by construction.  fp32: v_mfma_f32_* on fp32 operands.  The default run times "
                         "the other two as well and reports them in the same line.  bf16 (rounded operands) is not an "
                         "fp32 result and is labelled an experiment.")
    args = ap.parse_args()

    # `python bench.py --gpus N` on its own: start the N ranks as child processes (before anything here touches a GPU),
    # relay their output and leave with their status.  Under torch.distributed.run this is a no-op.
    from puresound_amd import launch
    if launch.needs_self_launch(args.gpus):
        raise SystemExit(launch.self_launch(os.path.abspath(__file__), sys.argv[1:], args.gpus))
    if launch.launch_probe("bench.py"):
        return

    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"bench.py --gpus {args.gpus} but WORLD_SIZE = {world}: launch with torch.distributed.run "
                         f"--nproc-per-node {args.gpus} (one rank per GPU)")
    dev = torch.device("cuda", local_rank)
    torch.cuda.set_device(dev)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)

    from puresound_amd import _abi
    from puresound_amd.batch_shard import OverlappedGather, gather_utterances, sharded_inference
    lib = _abi.lib()  # no HIP extension, no benchmark
    model = build_model(dev)
    model.masker.set_gemm_precision(args.gemm)
    if args.streams is not None:
        model.hip_streams = args.streams

    ragged = args.ragged and world > 1
    total_b = B_PER_GPU * world - (1 if ragged else 0)
    if ragged:
        # every rank holds the whole (synthetic) batch and runs its balanced contiguous share: the shards differ by one
        # utterance, the gather pads to the largest shard
        g = torch.Generator().manual_seed(1234)
        full = ((torch.rand(total_b, L, generator=g) * 2 - 1) * 0.5).to(dev)
        from puresound_amd.batch_shard import shard_bounds
        lo, hi = shard_bounds(total_b, world, rank)
        noisy = full[lo:hi]
    else:
        g = torch.Generator().manual_seed(1234 + rank)
        noisy = ((torch.rand(B_PER_GPU, L, generator=g) * 2 - 1) * 0.5).to(dev)  # synthetic 16 kHz waveforms

    # equal shards, N > 1: the all-gather of step i is issued asynchronously and runs (on RCCL's stream) under the
    # kernels of step i+1; fence() waits for every outstanding one, so all K gathers complete inside the timed region
    overlap = OverlappedGather(total_b) if (world > 1 and not ragged and not args.sync_gather) else None

    def step():
        if ragged:
            return sharded_inference(model.inference, full)
        out = model.inference(noisy)
        if overlap is not None:
            return overlap.submit(out)   # (the previous step's gathered batch)
        if world > 1:
            out = gather_utterances(out, total_b)
        return out

    def fence():
        if overlap is not None:
            overlap.flush()
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
    rank_ms = [elapsed / args.steps * 1e3]
    if world > 1:
        mine = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        every = torch.empty(world, device=dev, dtype=torch.float64)
        dist.all_gather_into_tensor(every, mine)
        rank_ms = [float(v) / args.steps * 1e3 for v in every.tolist()]
        elapsed = float(every.max().item())
    ms_per_step = elapsed / args.steps * 1e3
    value = total_b * L * args.steps / elapsed

    result = {
        "metric": "audio samples/sec (16 kHz) on ns Conv-TasNet, batch=32x4s per GPU",
        "value": value, "unit": "samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": {"fp32": "f32",
                  "fp16x2": "f32 (GEMM products issued as 2xfp16 split terms, 3 products, fp32 accumulate; measured error "
                            "against the reference equals the exact-fp32 MFMA path's)",
                  "bf16x3": "f32 (GEMM products issued as 3xbf16 split terms, fp32 accumulate; result error equals the "
                            "exact-fp32 MFMA path's)",
                  "bf16": "f32 storage/accumulate, bf16 products (EXPERIMENT, not an fp32 result)"}[args.gemm],
        "gemm": args.gemm,
        "data": "synthetic",
        "config": {"workload": "egs/ns Conv-TasNet, learned-conv encoder (32/16/512), R=3 X=8 H=256, "
                               "batch=32x4 s fp32 per GPU (BASELINE configs[1])",
                   "global_batch": total_b, "samples_per_utt": L, "parallelism": f"dp{world}",
                   "hip_streams_per_gpu": int(getattr(model, "hip_streams", 1)),
                   "shards": "ragged (balanced contiguous split of 32*N-1 utterances)" if ragged else "equal",
                   "x_realtime": value / SR},
        # what the collective layer actually saw (the driver's scaling run checks it against --gpus)
        "distributed": {"world_size": dist.get_world_size() if world > 1 else 1,
                        "backend": dist.get_backend() if world > 1 else None,
                        "collective": (None if world == 1 else
                                       "all_gather_into_tensor of [B/N, L] fp32 per step, asynchronous and double-buffered: "
                                       "step i's gather runs under step i+1's kernels; all complete inside the timed region"
                                       if overlap is not None else
                                       "all_gather_into_tensor of [B/N, L] fp32 inside the timed step"),
                        "ms_per_step_by_rank": rank_ms},
    }

    if rank == 0 and not args.no_roofline:
        # second pass over the same K steps with the library's per-launch hipEvents switched on.  The
        # sub-batch stream overlap is switched off for this pass: with two launches sharing the chip an
        # event pair would time the overlap, not the kernel.
        t = (L - 32) // 16 + 1
        streams_kept = getattr(model, "hip_streams", 1)
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
        c_ch, h_ch = model.masker.input_dim, model.masker.tcn_dim
        # algorithmic HBM bytes per launch of the HBM-bound kernels (fp32): dwconv reads and writes [32][H][T];
        # the encoder writes [32][C][T] (+ the waveform in); the decoder reads feats and mask [32][C][T] (+ waveform out)
        hbm_bytes = {"dwconv": 2.0 * B_PER_GPU * h_ch * t * 4,
                     "free_encode": B_PER_GPU * (c_ch * t + L) * 4.0,
                     "free_decode": B_PER_GPU * (2.0 * c_ch * t + L) * 4.0}
        for fam in ("dwconv", "free_encode", "free_decode") + (("absmax",) if args.gemm == "fp16x2" else ()):
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
        streams_note = ("kernel durations: hipEvents on the launch stream (ps_profile_enable) over the same K steps, one "
                        "stream; the timed region ran %d stream(s)" % int(streams_kept))
        common = {"bound": "mfma", "achieved": achieved, "unit": "TFLOP/s", "avg_launch_ms": avg_ms,
                  "note": streams_note,
                  "flop_per_launch": flops / launches, "launches_per_step": launches,
                  "kernel_ms_per_step": {k: v[0] / args.steps for k, v in fams.items()},
                  # HBM-bound kernels: algorithmic bytes / average launch duration, against the 8 TB/s peak
                  "hbm_bound_kernels": {k: {"avg_launch_ms": fams[k][0] / max(fams[k][1], 1),
                                            "algorithmic_bytes": hbm_bytes[k],
                                            "GBps": hbm_bytes[k] / (fams[k][0] / max(fams[k][1], 1) * 1e-3) / 1e9,
                                            "frac_of_8TBps": hbm_bytes[k] / (fams[k][0] / max(fams[k][1], 1) * 1e-3) / 8e12}
                                        for k in hbm_bytes if fams[k][1]}}
        if args.gemm == "fp32":
            traffic, src = pmc_traffic("fp32", "ps::conv1x1_")
            result["roofline"] = dict(common, peak=F32_MFMA_PEAK_TFLOPS, frac=achieved / F32_MFMA_PEAK_TFLOPS,
                                      traffic=traffic, kernel="ps::conv1x1_* (ps_conv1x1_f32)",
                                      traffic_note=f"HBM bytes per launch read from {src} (separate rocprofv3 --pmc "
                                                   f"passes of this command), not measured in this run" if src else
                                                   "no PMC summary committed for this arithmetic")
        else:
            planes_products = SPLIT_PRODUCTS[args.gemm]
            peak = BF16_MFMA_PEAK_TFLOPS / planes_products
            traffic, src = pmc_traffic(args.gemm, ("ps::conv1x1_bf16_", "ps::conv1x1_f16x2_"))
            # SURVEY 8(d), per TCN block and frame: in_conv reads C and writes H, the pointwise conv reads H and writes H,
            # out_conv reads H and the residual (C) and writes C: (3C + 4H) elements over three launches.  (out_conv's
            # two m-tiles each stream the same input rows; that second read is traffic, not algorithmic work.)
            alg_bytes = B_PER_GPU * t * 4.0 * (3 * c_ch + 4 * h_ch) / 3
            mfma_side = {"achieved_TFLOPs": achieved, "peak_TFLOPs": peak, "frac": achieved / peak,
                         "floor_ms": (flops / launches) / (peak * 1e12) * 1e3,
                         "note": f"algorithmic fp32 FLOP (2*M*K*T*N per launch) against the dense 16-bit MFMA peak "
                                 f"{BF16_MFMA_PEAK_TFLOPS:.0f} TFLOP/s / {planes_products} products per multiply-add; the "
                                 f"same FLOP against the fp32 MFMA peak {F32_MFMA_PEAK_TFLOPS} TFLOP/s = "
                                 f"{achieved / F32_MFMA_PEAK_TFLOPS:.2f}"}
            hbm_gbps = alg_bytes / (avg_ms * 1e-3) / 1e9
            hbm_side = {"achieved_GBps": hbm_gbps, "peak_GBps": HBM_PEAK_GBPS, "frac": hbm_gbps / HBM_PEAK_GBPS,
                        "floor_ms": alg_bytes / (HBM_PEAK_GBPS * 1e9) * 1e3}
            tnote = (f"HBM bytes per launch read from {src} (separate rocprofv3 --pmc passes of this command), not "
                     f"measured in this run" if src else "no PMC summary committed for this arithmetic")
            kern = ("ps::conv1x1_f16x2_rb_kernel (ps_conv1x1_f16x2_f32)" if args.gemm == "fp16x2" else
                    "ps::conv1x1_bf16_il_kernel (ps_conv1x1_bf16_f32)")
            # the binding roofline is the one with the larger floor for the average launch: the six-product split is
            # matrix-pipe bound (67 us against 60), the three-product split HBM bound (34 us against 60)
            if hbm_side["floor_ms"] > mfma_side["floor_ms"]:
                result["roofline"] = dict(common, bound="hbm", achieved=hbm_gbps, unit="GB/s", peak=HBM_PEAK_GBPS,
                                          frac=hbm_gbps / HBM_PEAK_GBPS, traffic=traffic, kernel=kern,
                                          algorithmic_bytes_per_launch=alg_bytes, mfma_side=mfma_side, traffic_note=tnote)
            else:
                result["roofline"] = dict(common, peak=peak, frac=achieved / peak, traffic=traffic, kernel=kern,
                                          peak_note=mfma_side["note"], algorithmic_bytes_per_launch=alg_bytes,
                                          hbm_side=hbm_side, traffic_note=tnote)
    if rank == 0 and "roofline" in result:
        # the whole step against HBM: SURVEY 8(d)'s algorithmic bytes of one forward (24 blocks x (3C + 6H) + 3C elements per
        # frame and utterance, + waveform in / out) / step time / 8 TB/s
        c_ch, h_ch = model.masker.input_dim, model.masker.tcn_dim
        blocks = model.masker.repeat_tcn * model.masker.per_tcn_stack
        t = (L - 32) // 16 + 1
        step_bytes = B_PER_GPU * (t * (blocks * (3 * c_ch + 6 * h_ch) + 3 * c_ch) + 2 * L) * 4.0
        result["roofline"]["step_algorithmic_bytes"] = step_bytes
        result["roofline"]["step_frac"] = step_bytes / (rank_ms[0] * 1e-3) / (HBM_PEAK_GBPS * 1e9)
    if rank == 0 and world == 1 and args.gemm in ("fp16x2", "bf16x3") and not args.no_roofline:
        # the other fp32-class arithmetics timed beside it, same inputs, same K steps: the exact-fp32 MFMA path
        # (v_mfma_f32_32x32x2_f32 on fp32 operands) and the six-product bf16 split
        def timed(gemm):
            model.masker.set_gemm_precision(gemm)
            for _ in range(max(2, args.warmup)):
                model.inference(noisy)
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(args.steps):
                model.inference(noisy)
            torch.cuda.synchronize(dev)
            dt = time.perf_counter() - t0
            model.masker.set_gemm_precision(args.gemm)
            return {"value": B_PER_GPU * L * args.steps / dt, "unit": "samples/s", "ms_per_step": dt / args.steps * 1e3}
        y_default = model.inference(noisy)
        result["fp32_mfma_path"] = dict(timed("fp32"), note="python bench.py --gemm fp32: the same step with v_mfma_f32 "
                                                            "on fp32 operands")
        # how far the timed arithmetic is from the exact-fp32 MFMA path on this very batch (unit-range waveforms)
        model.masker.set_gemm_precision("fp32")
        y_fp32 = model.inference(noisy)
        model.masker.set_gemm_precision(args.gemm)
        result["deviation_from_fp32_mfma_path"] = {
            "max_abs": float((y_default - y_fp32).abs().max()),
            "l2_rel": float((y_default - y_fp32).norm() / y_fp32.norm()),
            "note": "output waveforms of this batch, default arithmetic against --gemm fp32; against the reference's "
                    "golden vectors all three stand at 1e-6 (tests/test_fp16x2.py, tools/gemm_precision_error.py)"}
        other = "bf16x3" if args.gemm == "fp16x2" else "fp16x2"
        result[other + "_path"] = dict(timed(other), note=f"python bench.py --gemm {other}")
    if rank == 0 and world == 1 and args.gemm == "fp16x2" and not args.no_other_configs and not args.no_roofline:
        # BASELINE configs 3 / 4 / 5 are parity cases, not the metric -- but their timings should come from a run the driver
        # made, so a bounded pass of each rides along (tools/bench_configs.py; the same code tools/bench_recurrent.py runs)
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        import bench_configs as BC
        del model
        torch.cuda.empty_cache()
        other = {}
        for name, fn in (("cfg3", lambda: BC.cfg3(dev, steps=5)), ("cfg4", lambda: BC.cfg4(dev, steps=10)),
                         ("cfg5", lambda: BC.cfg5(dev, chunks=500)), ("ns_dpcrn", lambda: BC.ns_dpcrn(dev)),
                         ("ns_dparn", lambda: BC.ns_dparn(dev))):
            try:
                other[name] = fn()
            except Exception as e:  # a failure here must not take the headline line with it
                other[name] = {"error": f"{type(e).__name__}: {e}"}
            torch.cuda.empty_cache()
        result["other_configs"] = other
    if rank == 0 and not args.no_cpu_baseline and world == 1:
        result["cpu_baseline"] = cpu_baseline()
    if rank == 0:
        print(json.dumps(result), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
