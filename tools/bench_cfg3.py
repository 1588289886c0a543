"""BASELINE config 3 (td_tse_conv_tasnet_v0: 32 x (4 s mixture + 4 s enrolment)) in the three GEMM arithmetic modes."""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden"))
import json
import torch
import cases
from detweights import det_state_dict
import puresound_amd.nnet as PA
dev = "cuda:0"
model = cases.build(PA.NS, "cfg3_short").eval()
model.load_state_dict(det_state_dict(model))
model.to(dev)
g = torch.Generator().manual_seed(1234)
noisy = ((torch.rand(32, 64000, generator=g) * 2 - 1) * 0.5).to(dev)
enroll = ((torch.rand(32, 64000, generator=torch.Generator().manual_seed(1235)) * 2 - 1) * 0.5).to(dev)
ref = None
for prec in ("fp32", "bf16x3", "fp16x2", "bf16"):
    model.masker.set_gemm_precision(prec)
    for m in model.speaker_net:
        if hasattr(m, "gemm_precision"):
            m.gemm_precision = prec
    for _ in range(3):
        out = model.inference(noisy, enroll)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = model.inference(noisy, enroll)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 10 * 1e3
    if ref is None:
        ref = out.clone()
    l2 = float(torch.linalg.norm(out - ref) / torch.linalg.norm(ref))
    print(json.dumps({"config": "cfg3 td_tse_conv_tasnet_v0, 32 x (4 s + 4 s enrolment)", "gemm": prec, "ms_per_forward": ms,
                      "samples_per_s": 32 * 64000 / ms * 1e3, "l2_rel_vs_fp32": l2}), flush=True)
