"""Experiment: does running sub-batches on separate HIP streams de-phase the GEMM workgroups (GPU box)?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch, cases
import puresound_amd.nnet as PA
from puresound_amd import hip
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = cases.build(PA.NS, "cfg2_full").eval().to(dev)
noisy = ((torch.rand(32, 64000) * 2 - 1) * 0.5).to(dev)

def run(nsplit, reps=10):
    streams = [torch.cuda.Stream(dev) for _ in range(nsplit)]
    chunks = noisy.chunk(nsplit)
    blocks, nb = model.masker.block_array(dev)
    ws = [None] * nsplit
    def once():
        cur = torch.cuda.current_stream(dev)
        outs = []
        for i, (s, c) in enumerate(zip(streams, chunks)):
            s.wait_stream(cur)
            with torch.cuda.stream(s):
                feats, t = model.encoder.encode_padded(c)
                need = hip.lib().ps_conv_tasnet_workspace_bytes(c.shape[0], 512, 256, t)
                if ws[i] is None:
                    ws[i] = torch.zeros(need, dtype=torch.uint8, device=dev)
                mask = hip.conv_tasnet(blocks, nb, feats, t, 512, 256, None, False, ws[i])
                outs.append(model.encoder.decode_padded(feats, t, mask, "relu", "linear"))
        for s in streams:
            cur.wait_stream(s)
        return torch.cat(outs)
    for _ in range(3):
        out = once()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        out = once()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3, out

ref = None
for ns in (1, 2, 4, 8):
    ms, out = run(ns)
    if ref is None:
        ref = out
    print(f"streams={ns}: {ms:.2f} ms/step  {32*64000/ms/1e3:.1f} M samples/s  same={torch.equal(out, ref)}", flush=True)
