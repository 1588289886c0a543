"""hipGraph replay (puresound_amd.graphs.GraphedInference) of a few presets against their eager forward (GPU box): the new
launch paths of round 4 (frame-major recurrences, streamed-weight LSTM, LayerNorm epilogues, fp16x2 convolutions) inside a
captured graph.  One JSON line per preset; exits non-zero on a mismatch."""
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
from puresound_amd.graphs import GraphedInference  # noqa: E402


def main():
    dev = "cuda:0"
    bad = 0
    g = torch.Generator().manual_seed(5)
    noisy = ((torch.rand(32, 64000, generator=g) * 2 - 1) * 0.5).to(dev)
    enroll = ((torch.rand(32, 64000, generator=g) * 2 - 1) * 0.5).to(dev)
    for name in sys.argv[1:] or ("ns_dpcrn_short", "ns_dparn_short", "tse_skim_v2_short", "tse_unet_tcn_short", "cfg4_short"):
        model = cases.build(PA.NS, name).eval()
        model.load_state_dict(det_state_dict(model))
        model.to(dev)
        c = cases.CASES[name]
        spk = bool(c.get("speaker_net") or c.get("spk") or getattr(model, "embedding_free_tse", False))
        args = (noisy, enroll) if spk else (noisy,)
        torch.manual_seed(0)
        ref = model.inference(*args)
        fast = GraphedInference(model)
        torch.manual_seed(0)
        out = fast(*args)
        out = fast(*args)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(3):
            out = fast(*args)
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / 3 * 1e3
        same = bool(torch.equal(out, ref))
        err = float((out - ref).abs().max())
        ok = same or err < 1e-6
        bad += not ok
        print(json.dumps({"preset": name, "graph_ms": round(ms, 2), "bit_identical": same, "max_abs_diff": err, "ok": ok}), flush=True)
        del model, fast
        torch.cuda.empty_cache()
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
