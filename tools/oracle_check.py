"""HIP forward against the oracle (the CPU restatement, fp64) at sizes between the golden fixtures' and the benchmark's:
several 128-frame tiles with a partial last one, batch 2 (GPU box; the oracle is the checker here, as in tests/).
One JSON line per preset; exits non-zero above 1e-4 max-rel (the golden tests' tolerance)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests", "golden"), os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import cases  # noqa: E402
from detweights import det_state_dict, det_wave  # noqa: E402
from oracle import separator_oracle as O  # noqa: E402
import puresound_amd.nnet as PA  # noqa: E402


def rel_max(a, b):
    return float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))


def main():
    dev = "cuda:0"
    bad = 0
    names = sys.argv[1:] or ("ns_dpcrn_short", "ns_dparn_short", "tse_skim_causal_short", "tse_unet_tcn_short", "cfg4_short")
    for name in names:
        c = cases.CASES[name]
        model = cases.build(PA.NS, name).eval()
        sd = det_state_dict(model)
        model.load_state_dict(sd)
        model.to(dev)
        length = int(os.environ.get("PS_ORACLE_SAMPLES", "20000"))
        noisy = det_wave(901, 2, length)
        spk = "L_enroll" in c
        enroll = det_wave(902, 2, length) if spk else None
        row = {"preset": name, "samples": length}
        for prec in ("fp16x2", "fp32"):
            model.set_gemm_precision(prec)
            out = model.inference(noisy.to(dev), None if enroll is None else enroll.to(dev)).cpu()
            row["hip_" + prec] = out
        t0 = time.perf_counter()
        ref = O.inference(noisy.double(), {k: v.double() for k, v in sd.items()}, cases.oracle_cfg(name),
                          None if enroll is None else enroll.double())
        row["oracle_s"] = round(time.perf_counter() - t0, 1)
        edge = 16 if c["enc"]["kind"] == "stft" else 0
        sl = slice(edge, ref.shape[1] - edge) if edge else slice(None)
        for prec in ("fp16x2", "fp32"):
            row[prec] = rel_max(row.pop("hip_" + prec)[:, sl].numpy(), ref[:, sl].numpy())
        row["ok"] = row["fp16x2"] < 1e-4 and row["fp32"] < 1e-4
        bad += not row["ok"]
        print(json.dumps(row), flush=True)
        del model
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
