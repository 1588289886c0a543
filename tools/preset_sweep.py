"""Per-forward time of every egs preset mirror at 32 x 4 s on one GPU (GPU box): a sweep for outliers, not a benchmark.
  python tools/preset_sweep.py [name ...]      -> one JSON line per preset: ms in the exact-fp32 and fp16x2 arithmetics
Presets with a speaker branch get a 4 s enrolment per utterance."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import cases  # noqa: E402
from detweights import det_state_dict  # noqa: E402
import puresound_amd.nnet as PA  # noqa: E402

PRESETS = ["cfg1_short", "cfg2_short", "cfg3_short", "cfg3_causal_short", "cfg4_short", "cfg4_tse_short", "ns_dpcrn_short",
           "ns_dparn_short", "tse_unet_tcn_short", "tse_unet_tcn_causal_short", "tse_unet_tcn_v1_short", "tse_skim_v0_short",
           "tse_skim_v1_short", "tse_skim_v2_short", "tse_skim_causal_short", "tse_skim_fbank_short", "tse_skim_vad_short"]


def main():
    dev = "cuda:0"
    names = sys.argv[1:] or PRESETS
    g = torch.Generator().manual_seed(1234)
    noisy = ((torch.rand(32, 64000, generator=g) * 2 - 1) * 0.5).to(dev)
    enroll = ((torch.rand(32, 64000, generator=g) * 2 - 1) * 0.5).to(dev)
    for name in names:
        c = cases.CASES[name]
        try:
            model = cases.build(PA.NS, name).eval()
            model.load_state_dict(det_state_dict(model))
            model.to(dev)
            spk = bool(c.get("speaker_net") or c.get("spk") or getattr(model, "embedding_free_tse", False))
            fn = (lambda: model.inference(noisy, enroll)) if spk else (lambda: model.inference(noisy))
            out = {"preset": name}
            for prec in os.environ.get("PS_PRECS", "fp32,fp16x2").split(","):
                model.set_gemm_precision(prec)
                torch.manual_seed(0)
                for _ in range(2):
                    y = fn()
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(3):
                    y = fn()
                torch.cuda.synchronize()
                out[prec] = round((time.perf_counter() - t0) / 3 * 1e3, 2)
                out["finite"] = bool(torch.isfinite(y if torch.is_tensor(y) else y[0]).all())
            print(json.dumps(out), flush=True)
            del model
            torch.cuda.empty_cache()
        except Exception as e:  # keep sweeping
            print(json.dumps({"preset": name, "error": f"{type(e).__name__}: {e}"[:300]}), flush=True)


if __name__ == "__main__":
    main()
