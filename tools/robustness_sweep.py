"""fp16x2 (default) against exact-fp32 products for several presets at odd batch sizes and lengths (GPU box): every launch
eligibility rule (register-B GEMM, frame-major recurrences, LayerNorm epilogues, fp16x2 convolutions) flips somewhere in
this grid.  Prints one line per case; exits non-zero when an l2-rel exceeds 2e-5 or a result is not finite."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    sys.path.insert(0, p)
import cases  # noqa: E402
from detweights import det_state_dict  # noqa: E402
import puresound_amd.nnet as PA  # noqa: E402


def main():
    dev = "cuda:0"
    bad = 0
    for name in ("ns_dpcrn_short", "ns_dparn_short", "tse_skim_v2_short", "tse_skim_v1_short", "tse_unet_tcn_short", "cfg4_short"):
        model = cases.build(PA.NS, name).eval()
        model.load_state_dict(det_state_dict(model))
        model.to(dev)
        c = cases.CASES[name]
        spk = bool(c.get("speaker_net") or c.get("spk") or getattr(model, "embedding_free_tse", False))
        for n, length in ((1, 16000), (3, 40000), (8, 16000), (5, 64000), (16, 24000)):
            g = torch.Generator().manual_seed(n * 1000 + length)
            noisy = ((torch.rand(n, length, generator=g) * 2 - 1) * 0.5).to(dev)
            enroll = ((torch.rand(n, length, generator=g) * 2 - 1) * 0.5).to(dev) if spk else None
            outs = {}
            for prec in ("fp32", "fp16x2"):
                model.set_gemm_precision(prec)
                torch.manual_seed(0)
                outs[prec] = model.inference(noisy, enroll) if spk else model.inference(noisy)
            err = float(torch.linalg.norm(outs["fp16x2"] - outs["fp32"]) / torch.linalg.norm(outs["fp32"]))
            ok = bool(torch.isfinite(outs["fp16x2"]).all()) and err < 2e-5
            bad += not ok
            print(json.dumps({"preset": name, "batch": n, "samples": length, "l2_rel": err, "ok": ok}), flush=True)
        del model
        torch.cuda.empty_cache()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
