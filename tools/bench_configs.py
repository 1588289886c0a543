"""Timed lines for BASELINE configs 3, 4 and 5 (bench.py's `other_configs`, tools/bench_recurrent.py, tools/bench_cfg3.py).
Every function builds the named configuration with the deterministic weights, runs on the given device and returns a
dict of plain numbers; nothing here prints."""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

L = 64000


def _waves(n, seed, dev):
    g = torch.Generator().manual_seed(seed)
    return ((torch.rand(n, L, generator=g) * 2 - 1) * 0.5).to(dev)


def _build(name, dev):
    import cases
    from detweights import det_state_dict
    import puresound_amd.nnet as PA
    model = cases.build(PA.NS, name).eval()
    model.load_state_dict(det_state_dict(model))
    return model.to(dev)


def _timed(fn, steps, warmup):
    for _ in range(warmup):
        out = fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / steps * 1e3, out


def cfg3(dev, steps=5, warmup=2, modes=("fp16x2", "bf16"), batch=32):
    """td_tse_conv_tasnet_v0 (egs/tse/model.py:95-140), batch x (4 s mixture + 4 s enrolment).  "bf16" is the arithmetic
    BASELINE names for this config (bf16 products and hidden rows, fp32 accumulation); l2_rel is taken against an
    exact-fp32 run of the same batch."""
    model = _build("cfg3_short", dev)
    model.hip_streams = 1   # (one stream: every launch covers the whole batch, as in bench.py's timed path; two lanes: 14.2 ms)
    noisy, enroll = _waves(batch, 1234, dev), _waves(batch, 1235, dev)

    def set_mode(prec):
        model.masker.set_gemm_precision(prec)
        for m in model.speaker_net:
            if hasattr(m, "gemm_precision"):
                m.gemm_precision = prec
    set_mode("fp32")
    ref = model.inference(noisy, enroll)
    out = {"workload": f"td_tse_conv_tasnet_v0, {batch} x (4 s mixture + 4 s enrolment), 1 GPU", "steps": steps}
    # 0.745 GB per utterance pair in bf16 storage (mixture 24 blocks + enrolment 5 blocks + features), twice
    # that with fp32 rows; roofline_frac = those bytes / time / 8 TB/s
    alg = {"bf16": 23.86e9 * batch / 32, "fp16x2": 47.72e9 * batch / 32, "bf16x3": 47.72e9 * batch / 32, "fp32": 47.72e9 * batch / 32}
    for prec in modes:
        set_mode(prec)
        ms, y = _timed(lambda: model.inference(noisy, enroll), steps, warmup)
        out[prec] = {"ms": ms, "samples_s": batch * L / ms * 1e3,
                     "l2_rel_vs_fp32": float(torch.linalg.norm(y - ref) / torch.linalg.norm(ref)),
                     "algorithmic_bytes": alg[prec], "roofline_frac": alg[prec] / (ms * 1e-3) / 8e12}
    if "fp16x2" in modes:
        # td_tse_conv_tasnet_v0_causal (egs/tse/model.py:142-183: bN1d blocks, causal): same bytes, the fp16x2 ranges behind
        # the folded BatchNorms come from maxima the producing kernels measure
        del model
        causal = _build("cfg3_causal_short", dev)
        causal.hip_streams = 1
        causal.set_gemm_precision("fp32")
        ref = causal.inference(noisy, enroll)
        causal.set_gemm_precision("fp16x2")
        ms, y = _timed(lambda: causal.inference(noisy, enroll), steps, warmup)
        out["causal_bn_fp16x2"] = {"workload": "td_tse_conv_tasnet_v0_causal", "ms": ms, "samples_s": batch * L / ms * 1e3,
                                   "l2_rel_vs_fp32": float(torch.linalg.norm(y - ref) / torch.linalg.norm(ref)),
                                   "algorithmic_bytes": alg["fp16x2"], "roofline_frac": alg["fp16x2"] / (ms * 1e-3) / 8e12}
    return out


def cfg4_model(dev, gemm="fp32"):
    model = _build("cfg4_short", dev)
    model.masker.set_gemm_precision(gemm)
    model.hip_streams = int(os.environ.get("PS_CFG4_STREAMS", "1"))
    return model


def cfg4(dev, steps=10, warmup=3, batch=32, gemm="fp32", graph=True):
    """FreeEncDec(32,16,128) + DPRNN(128,64,128, 6 blocks, K=20, causal), batch x 4 s per GPU: eager and as one hipGraph.
    serial_steps = the 6 x (20 + 200) dependent LSTM cell steps of one forward."""
    model = cfg4_model(dev, gemm)
    noisy = _waves(batch, 1234, dev)
    ms, _ = _timed(lambda: model.inference(noisy), steps, warmup)
    out = {"workload": f"DPRNN(128,64,128,6 blocks,K=20,causal), {batch} x 4 s, 1 GPU, fp32 rows, input projections {gemm}",
           "steps": steps, "ms": ms, "samples_s": batch * L / ms * 1e3, "serial_steps": 6 * (20 + 200),
           "us_per_serial_step": ms * 1e3 / (6 * (20 + 200))}
    # (6 * 4C + 4C) elements per frame, T' = 4000 frames, fp32 rows here (bf16 storage: half); the config is
    # bounded by its 1320 dependent LSTM steps, so the HBM fraction rides along with the microseconds per step
    alg = (6 * 4 * 128 + 4 * 128) * 4000 * 4.0 * batch
    out["algorithmic_bytes"] = alg
    out["roofline_frac"] = alg / (ms * 1e-3) / 8e12
    if graph:
        from puresound_amd.graphs import GraphedInference
        fast = GraphedInference(model)
        msg, _ = _timed(lambda: fast(noisy), steps, 3)
        out["hipgraph_ms"] = msg
        out["hipgraph_us_per_serial_step"] = msg * 1e3 / (6 * (20 + 200))
    if gemm == "fp32":
        # the same forward with the masker in the fp16x2 arithmetic (fp32-class: l2_rel against the run above)
        ref = model.inference(noisy)
        model.masker.set_gemm_precision("fp16x2")
        ms2, y = _timed(lambda: model.inference(noisy), steps, warmup)
        out["fp16x2_projections"] = {"ms": ms2, "samples_s": batch * L / ms2 * 1e3, "roofline_frac": alg / (ms2 * 1e-3) / 8e12,
                                     "l2_rel_vs_fp32": float(torch.linalg.norm(y - ref) / torch.linalg.norm(ref)),
                                     "arithmetic": "masker.set_gemm_precision('fp16x2'): LSTM input projections and the "
                                                   "intra-pass recurrent product W_hh h in two fp16 terms per operand"}
        model.masker.set_gemm_precision("fp32")
    return out


def ns_dparn(dev, steps=3, warmup=2, batch=32, gemm="fp16x2"):
    """egs/ns/model.py:128-171 (ns_dparn_v0_causal: the DPCRN with self-attention along frequency, nhead = 8)."""
    return ns_dpcrn(dev, steps, warmup, batch, gemm, case="ns_dparn_short", name="ns_dparn_v0_causal")


def ns_dpcrn(dev, steps=3, warmup=2, batch=32, gemm="fp16x2", case="ns_dpcrn_short", name="ns_dpcrn_v0_causal"):
    """The real egs/ns model (ns_dpcrn_v0_causal, egs/ns/model.py:40-82: conv-STFT 512/128 + DPCRN(1,32,32,32,64,128; H=128)
    + complex mask + iSTFT), batch x 4 s; `gemm` = arithmetic of the LSTM input projections and linear layers."""
    model = _build(case, dev)
    noisy = _waves(batch, 1234, dev)
    out = {"workload": f"{name}, {batch} x 4 s, 1 GPU, fp32 rows", "steps": steps}
    model.masker.set_gemm_precision("fp32")
    ref = model.inference(noisy)
    # algorithmic bytes (fp32 rows, every layer reading its input and writing its output once; T = 501 frames): the ten
    # convolutions move 87,040 (channel x frequency) rows per frame, the four recurrent passes read and write the 128 x 64
    # bottleneck map once each; pre-activations, h' and the STFT are fusible and not counted
    alg = (87040 + 4 * 2 * 8192) * 501 * 4.0 * batch
    out["algorithmic_bytes"] = alg if case == "ns_dpcrn_short" else None
    for prec in ("fp32", gemm):
        model.masker.set_gemm_precision(prec)
        ms, y = _timed(lambda: model.inference(noisy), steps, warmup)
        out[prec] = {"ms": ms, "samples_s": batch * L / ms * 1e3, "roofline_frac": alg / (ms * 1e-3) / 8e12,
                     "l2_rel_vs_fp32": float(torch.linalg.norm(y - ref) / torch.linalg.norm(ref))}
        if case != "ns_dpcrn_short":   # (the byte count above is DPCRN's; the attention variant's q / k / v rows are fusible too)
            out[prec]["roofline_frac"] = None
    out["roofline_note"] = ("bounded by the fp32 matrix pipe in the convolutions (651 GFLOP at 157 TFLOP/s = 4.1 ms) and by the "
                            "6.4 GB round trip of the LSTM gate pre-activations, not by the algorithmic bytes")
    return out


def cfg5(dev, chunks=500, streams=64, warmup=10):
    """Demo preset (egs/tse/demo/utils.py:51-72), `streams` concurrent streams, 320-sample chunks, one hipGraph per chunk."""
    from detweights import det_state_dict
    from puresound_amd.streaming.demo import DemoTseNet
    net = DemoTseNet().eval()
    net.load_state_dict(det_state_dict(net))
    net.to(dev)
    net.init_streams(streams)
    g = torch.Generator().manual_seed(1236)
    embed = torch.rand(streams, 192, generator=g).to(dev)
    wav = ((torch.rand(streams, 320 * 8, generator=g) * 2 - 1) * 0.5).to(dev)
    lat, pre = [], None
    for i in range(chunks + warmup):
        chunk = wav[:, (i % 8) * 320:(i % 8 + 1) * 320]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        y = net.streaming_inference_chunk(chunk, embed, pre)
        pre = y[:, -16:]
        torch.cuda.synchronize()
        if i >= warmup:
            lat.append((time.perf_counter() - t0) * 1e3)
    lat = np.array(lat)
    return {"workload": f"demo StreamingSkiM(128,256,128,4 blocks,K=150), {streams} streams x 320-sample chunks, one hipGraph "
                        f"per chunk", "chunks": int(len(lat)), "p50_ms": float(np.percentile(lat, 50)),
            "p90_ms": float(np.percentile(lat, 90)), "max_ms": float(lat.max()), "budget_ms": 20.0,
            "mem_lstm_updates_seen": int(len(lat) * 20 // 150),
            "roofline_frac": None, "roofline_note": "latency-bound (SURVEY 8d): 80 dependent LSTM steps per chunk against "
                                                    "a 20 ms budget; no bandwidth roofline applies"}
